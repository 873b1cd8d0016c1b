#!/usr/bin/env python3
"""bench.py -- Msamples/s (paths x bounces = RayTracingCounters::numRays) of the MI355X PathTracerMIS core.

Workload (BASELINE.json configs[2]): Sponza-class triangle mesh (~262k triangles, procedural stand-in: the
reference's sponza.obj is not shipped), PathTracerMIS, 8 bounces, 1920x1080, background + delta directional
light, LightSamplingStrategy::Single.  One "step" = one pass = one sample per pixel of the whole frame.

  python bench.py --gpus 1 --steps 256 --warmup 16        (the defaults: BASELINE config 3 renders 256 samples per pixel)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1: the frame's 64x64 tiles are interleaved across ranks (tile % N == rank, identical scene on every GPU, no
data-path collective); after the K timed passes the float3 sum buffers (disjoint support) are sum-reduced to
rank 0 over RCCL, inside the timed region.  Total work is fixed => "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)   # the 256 spp of BASELINE config 3: the whole frame of the metric's configuration
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--triangles", type=int, default=262144)
    ap.add_argument("--workload", default="sponza", choices=["sponza", "sponza-textured", "cornell", "sphere"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the bounded CPU-baseline sample")
    return ap.parse_args()


def build_scene(args, aspect):
    from raytracer_amd import scenes
    if args.workload in ("sponza", "sponza-textured"):
        return scenes.sponza_class(aspect, args.triangles, textured=args.workload == "sponza-textured")
    if args.workload == "cornell":
        return scenes.cornell_box(aspect)
    return scenes.sphere_area_light(aspect)


def device_tensor(ptr, num_floats, torch):
    """torch view of a hipMalloc'ed float buffer owned by librtgpu (plumbing for the RCCL reduce)."""
    class _Buf:
        pass
    b = _Buf()
    b.__cuda_array_interface__ = {"shape": (int(num_floats),), "typestr": "<f4", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(b, device="cuda")


def algorithmic_bytes(c):
    """per interior-node visit two 32-byte children, 36 bytes per triangle test, 176 B per mesh hit,
    192 B per analytic hit, 24 B film read-modify-write per path (+12 B on even passes => x1.5)."""
    trace_closest = 32 * c["numRayBoxTests"] + 36 * c["numRayTriangleTests"]
    trace_shadow = 32 * c["numShadowRayBoxTests"] + 36 * c["numShadowRayTriangleTests"]
    shade = 176 * c["numMeshHits"] + 192 * c["numAnalyticHits"]
    film = 36 * c["numPrimaryRays"]
    return {"trace": trace_closest + trace_shadow, "shade": shade, "accumulate": film, "generate": 0}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; BENCH_DIST_BACKEND=gloo with ranks sharing a device exists only to exercise the N>1 code path
    # on a 1-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        dist.barrier()
    import raytracer_amd as ra

    w, h = args.width, args.height
    scene, camera = build_scene(args, w / h)
    vp = ra.Viewport(w, h, seed=20260928, max_ray_depth=args.depth)
    vp.set_renderer(scene, device=local_rank)
    if world > 1:
        vp.set_shard(rank, world)
    elif os.environ.get("BENCH_EMULATE_SHARD"):
        # tuning aid on a 1-GPU box: render only the tiles rank 0 of N would own ("value" is then that rank's share)
        vp.set_shard(0, int(os.environ["BENCH_EMULATE_SHARD"]))
    lib = ra.rtgpu_lib()
    ctx = vp.device_context()

    def sync_all():
        lib.rtgpu_synchronize(ctx)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # every rank draws the same per-pass constants (same seed => same Halton / AA offsets).
    # The box / triangle test counters are instrumentation (a compile-time debug switch in the reference,
    # RT_ENABLE_INTERSECTION_COUNTERS, off by default): they are OFF in the timed region and collected afterwards
    # by replaying exactly the same passes (same seed => same rays) with the counters on.
    lib.rtgpu_set_intersection_counters(ctx, 0)
    vp.render(camera, args.warmup)
    sync_all()
    c0 = vp.counters()
    lib.rtgpu_enable_timing(ctx, 1)

    sum_ptr, sec_ptr, nfl = C.c_void_p(), C.c_void_p(), C.c_size_t()
    lib.rtgpu_get_device_sum(ctx, C.byref(sum_ptr), C.byref(sec_ptr), C.byref(nfl))

    sync_all()
    t0 = time.perf_counter()
    vp.render(camera, args.steps)
    lib.rtgpu_synchronize(ctx)
    if world > 1:
        t_sum = device_tensor(sum_ptr.value, nfl.value, torch)
        dist.reduce(t_sum, dst=0, op=dist.ReduceOp.SUM)   # disjoint tile support => exact gather
    sync_all()
    elapsed = time.perf_counter() - t0

    c1 = vp.counters()
    delta = {k: c1[k] - c0[k] for k in c1}
    lib.rtgpu_enable_timing(ctx, 0)

    def kernel_times(context):
        ms = (C.c_double * 8)(); launches = (C.c_uint64 * 8)(); names = (C.c_char_p * 8)()
        lib.rtgpu_get_kernel_times(context, ms, launches, names)
        return {names[i].decode(): (ms[i], int(launches[i])) for i in range(8) if names[i]}

    def replay(lanes, intersection_counters, timing):
        """Renders exactly the same passes again (same seed => same rays) on a fresh viewport; returns the counter
        deltas of the `steps` passes, the counter totals of warm-up + steps and, if asked, the per-kernel-class HIP-event
        times of warm-up + steps (every launch of the replay: the population `rocprofv3 --kernel-trace --stats` averages over
        when the same command runs with RTGPU_LANES=1)."""
        v = ra.Viewport(w, h, seed=20260928, max_ray_depth=args.depth)
        v.set_renderer(scene, device=local_rank)
        if world > 1:
            v.set_shard(rank, world)
        elif os.environ.get("BENCH_EMULATE_SHARD"):
            v.set_shard(0, int(os.environ["BENCH_EMULATE_SHARD"]))
        vctx = v.device_context()
        lib.rtgpu_set_concurrency(vctx, lanes)
        lib.rtgpu_set_intersection_counters(vctx, 1 if intersection_counters else 0)
        lib.rtgpu_enable_timing(vctx, 1 if timing else 0)
        v.render(camera, args.warmup)
        a0 = v.counters()
        v.render(camera, args.steps)
        a1 = v.counters()
        times = kernel_times(vctx) if timing else None
        return {k: a1[k] - a0[k] for k in a1}, dict(a1), times

    # kernel-class times measured with HIP events on the streams the kernels run on, over the timed region: with
    # several batch lanes the kernels of consecutive batches OVERLAP, so these durations include time shared with
    # another kernel.  The roofline therefore uses a lanes=1 replay of the same passes (kernels strictly serial).
    ktimes_overlapped = kernel_times(ctx)
    serial, _, ktimes = replay(1, False, True)
    assert serial["numRays"] == delta["numRays"] and serial["numShadowRays"] == delta["numShadowRays"], "serial replay diverged from the timed run"
    # instrumented replay for the intersection counters (not timed)
    counted, counted_totals, _ = replay(1, True, False)
    assert counted["numRays"] == delta["numRays"] and counted["numShadowRays"] == delta["numShadowRays"], "replay diverged from the timed run"
    for k in ("numRayBoxTests", "numPassedRayBoxTests", "numRayTriangleTests", "numPassedRayTriangleTests", "numShadowRayBoxTests",
              "numShadowRayTriangleTests"):
        delta[k] = counted[k]
    own_counts = dict(delta)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        keys = sorted(delta)
        tc = torch.tensor([delta[k] for k in keys], dtype=torch.int64, device="cuda")
        dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        delta = {k: int(v) for k, v in zip(keys, tc.tolist())}

    if rank == 0:
        value = delta["numRays"] / elapsed / 1e6
        out = {
            "metric": "Msamples/sec (paths x bounces) at %dx%d, Sponza-class PT-MIS" % (w, h),
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / max(1, args.steps), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: %s, PathTracerMIS, %d bounces, %dx%d, LightSamplingStrategy::Single" %
                       ("procedural Sponza-class mesh (%d triangles, 8 diffuse materials)" % scene.desc.contents.numTriangles
                        if args.workload.startswith("sponza") else args.workload, args.depth, w, h)
                       + (" + albedo / normal maps on all materials, HDR environment map" if args.workload == "sponza-textured" else ""),
                       "spp_timed": args.steps, "parallelism": "tile-interleaved x%d" % world},
            "counters": {k: delta[k] for k in ("numRays", "numPrimaryRays", "numShadowRays", "numRayBoxTests", "numRayTriangleTests",
                                               "numShadowRayBoxTests", "numShadowRayTriangleTests", "numMeshHits", "numAnalyticHits")},
            "mrays_per_s_incl_shadow": (delta["numRays"] + delta["numShadowRays"]) / elapsed / 1e6,
            "intersection_counters": "off in the timed region (reference default); counts from an identical instrumented replay",
        }
        # roofline of the dominant kernel class (rank 0's own launches and rank 0's own counters)
        abytes = algorithmic_bytes(own_counts)
        abytes_replay = algorithmic_bytes(counted_totals)   # warm-up + timed passes: what the replay's launches processed
        dom = max(ktimes, key=lambda k: ktimes[k][0]) if ktimes else None
        if dom and ktimes[dom][0] > 0:
            per_launch_bytes = abytes_replay[dom] / max(1, ktimes[dom][1])
            per_launch_s = ktimes[dom][0] / 1000.0 / max(1, ktimes[dom][1])
            achieved = per_launch_bytes / per_launch_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch_upper")
                except Exception:
                    traffic = None
            out["roofline"] = {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": per_launch_s * 1000.0,
                               "launches": ktimes[dom][1],
                               "measured": "HIP events around every launch of a lanes=1 replay of the warm-up and timed passes (serial "
                                           "kernels; batches growing 2, 4, 8, 16, then 24 passes per launch); the timed region itself runs 3 batch "
                                           "lanes whose kernels overlap"}
            out["kernel_time_ms"] = {k: round(v[0], 3) for k, v in ktimes.items()}
            out["kernel_time_ms_timed_region_overlapped"] = {k: round(v[0], 3) for k, v in ktimes_overlapped.items()}
            tot_bytes = sum(abytes.values())
            out["whole_pass_algorithmic_GBs"] = tot_bytes / elapsed / 1e9

        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, scene, camera, ra)
        line = json.dumps(out)
    else:
        line = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(line, flush=True)   # after the process group is gone: nothing the communication library prints can follow it


def cpu_baseline(args, scene, camera, ra):
    """The scalar CPU oracle ("port" of the reference algorithm -- the reference renderer itself cannot be built
    in this image) timed on all host cores on a BOUNDED sample of the same workload: rows of the
    same frame, same scene, same depth, as many whole passes as fit the time budget (at least a band of one)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    w, h = args.width, args.height
    cores = os.cpu_count() or 1
    desc = scene.desc
    bn = ra.load_blue_noise()
    desc.contents.blueNoise = bn.ctypes.data
    vp = ra.Viewport(w, h, seed=77, max_ray_depth=args.depth)
    img = np.zeros((h, w, 3), dtype=np.float32)
    cnt = np.zeros(16, dtype=np.uint64)
    # calibration on a thin band (every 16th 64x64 tile), then whole passes within the budget
    p = vp.next_pass_params(camera)
    t0 = time.perf_counter()
    oracle_lib.render_pass(desc, p, w, h, img, None, cnt, shard=(0, 16), threads=cores)
    t_band = time.perf_counter() - t0
    rays_band = int(cnt[0])
    rate = rays_band / t_band
    sample = "1/16 of the tiles of one pass"
    total_rays, total_t = rays_band, t_band
    est_pass = t_band * 16.0
    passes = int(max(0.0, args.cpu_seconds - t_band) // max(est_pass, 1e-9))
    if passes >= 1:
        passes = min(passes, 4)
        cnt[:] = 0
        t0 = time.perf_counter()
        for _ in range(passes):
            p = vp.next_pass_params(camera)
            oracle_lib.render_pass(desc, p, w, h, img, None, cnt, threads=cores)
        total_t = time.perf_counter() - t0
        total_rays = int(cnt[0])
        rate = total_rays / total_t
        sample = "%d full pass(es) of the %dx%d frame" % (passes, w, h)
    return {"value": rate / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port", "sample": sample,
            "seconds": round(total_t, 2), "numRays": total_rays}


if __name__ == "__main__":
    main()
