"""Throughput of the bidirectional integrator (renderer "VCM") on a caustics scene in the spirit of BASELINE config 5
(rough-glass dielectric object lit through by a small area light, 1920x1080, path length 10).  Not the headline bench
(bench.py keeps the PathTracerMIS metric); prints one JSON line: passes/s, path segments (numRays) per second, shadow
rays per second, and the per-kernel-class times of the pass.

    python tools/bench_vcm.py [--width 1920 --height 1080 --passes 8 --warmup 2 --scene caustics|sponza]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def caustics_scene(ra, aspect):
    s = ra.Scene()
    floor = s.add_material("diffuse", (0.75, 0.75, 0.75))
    red = s.add_material("diffuse", (0.75, 0.2, 0.2))
    green = s.add_material("diffuse", (0.2, 0.75, 0.2))
    glass = s.add_material("roughDielectric", (1.0, 1.0, 1.0), roughness=0.08, ior=1.5)
    metal = s.add_material("roughMetal", (0.9, 0.8, 0.6), roughness=0.2)
    s.add_rect((6.0, 6.0), ra.transform_from_euler((0.0, 0.0, 0.0), (-90.0, 0.0, 0.0)), floor)
    s.add_rect((6.0, 3.0), ra.transform_from_euler((0.0, 3.0, -6.0), (0.0, 0.0, 0.0)), floor)
    s.add_rect((6.0, 3.0), ra.transform_from_euler((-6.0, 3.0, 0.0), (0.0, 90.0, 0.0)), red)
    s.add_rect((6.0, 3.0), ra.transform_from_euler((6.0, 3.0, 0.0), (0.0, -90.0, 0.0)), green)
    s.add_sphere(1.2, ra.transform_from_euler((-1.8, 1.2, -0.5)), glass)
    s.add_box((0.9, 1.4, 0.9), ra.transform_from_euler((2.2, 1.4, -1.5), (0.0, 25.0, 0.0)), glass)
    s.add_sphere(0.7, ra.transform_from_euler((0.6, 0.7, 1.8)), metal)
    s.add_point_light((60.0, 55.0, 50.0), ra.transform_from_euler((-3.5, 5.0, 2.5)))
    s.add_spot_light((300.0, 300.0, 280.0), 0.35, ra.transform_from_euler((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)))
    s.add_background_light((0.05, 0.07, 0.1))
    s.build()
    return s, ra.Camera((0.0, 3.0, 9.5), (12.0, 180.0, 0.0), aspect, 50.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--passes", type=int, default=8); ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="caustics"); ap.add_argument("--max-path-length", type=int, default=10)
    ap.add_argument("--no-merging", action="store_true"); ap.add_argument("--no-connection", action="store_true")
    args = ap.parse_args()
    import __graft_entry__ as entry
    entry.build()
    import raytracer_amd as ra
    from raytracer_amd import scenes
    w, h = args.width, args.height
    if args.scene == "sponza":
        scene, camera = scenes.sponza_class(w / h)
    else:
        scene, camera = caustics_scene(ra, w / h)
    vp = ra.Viewport(w, h, seed=20260928)
    vp.set_renderer(scene, name="VCM")
    vp.set_vcm(max_path_length=args.max_path_length, use_vertex_merging=not args.no_merging, use_vertex_connection=not args.no_connection)
    lib = ra.rtgpu_lib(); ctx = vp.device_context()
    lib.rtgpu_set_intersection_counters(ctx, 0)
    vp.render(camera, args.warmup)
    lib.rtgpu_synchronize(ctx)
    c0 = vp.counters()
    lib.rtgpu_enable_timing(ctx, 1)
    t0 = time.perf_counter()
    vp.render(camera, args.passes)
    lib.rtgpu_synchronize(ctx)
    dt = time.perf_counter() - t0
    c1 = vp.counters()
    times = (C.c_double * 8)(); launches = (C.c_uint64 * 8)()
    names = (C.c_char_p * 8)()
    lib.rtgpu_get_kernel_times(ctx, times, launches, names)
    img = vp.sum_buffer()
    d = {k: c1[k] - c0[k] for k in c1}
    print(json.dumps({"renderer": "VCM", "scene": args.scene, "width": w, "height": h, "passes": args.passes, "max_path_length": args.max_path_length,
                      "ms_per_pass": 1e3 * dt / args.passes, "msegments_per_s": d["numRays"] / dt / 1e6, "mshadow_rays_per_s": d["numShadowRays"] / dt / 1e6,
                      "photons_last_pass": vp.vcm_num_photons(), "kernel_ms": {n: round(times[i], 3) for i, n in enumerate(("generate", "trace", "shade", "accumulate"))},
                      "image_mean": float(img.mean() / (args.passes + args.warmup)), "finite": bool(np.isfinite(img).all())}))


if __name__ == "__main__":
    main()
