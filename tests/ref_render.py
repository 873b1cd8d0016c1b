"""Driver of oracle/_ref/ref_render -- the REFERENCE'S OWN AVX2 / FMA object code (Viewport::Render -> ThreadPool ->
PathTracerMIS::RenderPixel -> traversal / intersection / shading; see oracle/ref_harness/ref_render.cpp for what is glue) behind a small
scene file.  TEST / MEASUREMENT INFRASTRUCTURE: used by tests (image-level statistics of the real integrator) and by bench.py's
`cpu_baseline` leg; the product package never imports it.

export_scene() writes what raytracer_amd.Scene recorded in `.calls` (the very inputs the GPU scene was built from) as plain arrays; the
binary rebuilds the scene through the reference's public API (MeshShape::Initialize builds its own BVH from the same vertices)."""
import json
import os
import shutil
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_render")
# the reference opens "../Data/BlueNoise128_RGBA16.dat" relative to its working directory (Core/Sampling/GenericSampler.cpp:13)
DATA_PARENT = os.path.join(ROOT, "raytracer_amd")

BSDF_IDS = {"null": 0, "diffuse": 1, "roughDiffuse": 2, "dielectric": 3, "roughDielectric": 4, "metal": 5, "roughMetal": 6, "plastic": 7, "roughPlastic": 8}


def available():
    return os.path.exists(EXE) and os.access(EXE, os.X_OK)


def export_scene(path, scene, camera, width, height, passes, threads, max_ray_depth, min_rr_depth=1, dimensions=64, use_blue_noise=True,
                 light_sampling_all=False, aa_spread=0.5, seed=1234, dump_image=True):
    materials, meshes, objects, lights, textures = [], [], [], [], []
    texture_index = {}     # the library's texture id -> index in the exported list
    slots = dict(baseColor=0, emission=1, roughness=2, metalness=3, normal=4)
    material_textures = {}  # material -> [five texture indices or -1, normalMapStrength]
    for kind, a in scene.calls:
        if kind == "bitmap_texture":
            # the reference's public API constructs a BitmapTexture with its default filter and no palette: only those are exported
            if a["filter"] != "smoothstep" or a["has_palette"] or a["stride"] == 0:
                raise ValueError("only plain bitmaps with the default filter are exported")
            raw = a["pixels"].tobytes()
            texture_index[a["id"]] = len(textures)
            textures.append(struct.pack("<8I", a["width"], a["height"], scene.FORMATS[a["format"]], int(a["linear_space"]), a["stride"], len(raw), 0, 0) + raw + b"\0" * (-len(raw) % 4))
        elif kind == "material_texture":
            entry = material_textures.setdefault(a["material"], [-1, -1, -1, -1, -1, 1.0])
            entry[slots[a["slot"]]] = texture_index[a["texture"]]
            if a["slot"] == "normal":
                entry[5] = a["strength"]
    material_count = 0
    for kind, a in scene.calls:
        if kind in ("bitmap_texture", "material_texture"):
            continue
        if kind == "material":
            t = material_textures.get(material_count, [-1, -1, -1, -1, -1, 1.0])
            material_count += 1
            materials.append(struct.pack("<I10f5if", BSDF_IDS[a["bsdf"]], *a["base_color"], *a["emission"], a["roughness"], a["metalness"], a["ior"], a["k"], *t[:5], t[5]))
        elif kind in ("sphere", "box", "rect"):
            p = {"sphere": [a.get("radius", 0.0), 0, 0, 0], "box": list(a.get("size", (0, 0, 0))) + [0], "rect": list(a.get("size", (0, 0))) + list(a.get("tex_scale", (1, 1)))}[kind]
            objects.append(struct.pack("<IiII4f16f", {"sphere": 0, "box": 1, "rect": 2}[kind], a["material"], 0, 0, *p, *a["transform"]))
        elif kind == "mesh":
            pos, idx = a["positions"], a["indices"]
            nv, nt = pos.shape[0], idx.shape[0]
            zeros3, zeros2 = np.zeros((nv, 3), np.float32), np.zeros((nv, 2), np.float32)
            nrm = a["normals"] if a["normals"] is not None else zeros3
            tan = a["tangents"] if a["tangents"] is not None else zeros3
            uv = a["tex_coords"] if a["tex_coords"] is not None else zeros2
            table = list(a["materials"]) or [max(a["material"], 0)]
            mi = a["material_indices"] if a["material_indices"] is not None else np.zeros(nt, np.uint32)
            blob = struct.pack("<4I", nv, nt, len(table), 0) + pos.astype("<f4").tobytes() + nrm.astype("<f4").tobytes() + tan.astype("<f4").tobytes() + \
                uv.astype("<f4").tobytes() + idx.astype("<u4").tobytes() + np.asarray(mi, "<u4").tobytes() + np.asarray(table, "<u4").tobytes()
            objects.append(struct.pack("<IiII4f16f", 3, a["material"], len(meshes), 0, 0, 0, 0, 0, *a["transform"]))
            meshes.append(blob)
        elif kind == "area_light":
            params = list(a["params"])
            if a["shape"] == 2:
                params[2:4] = [1.0, 1.0]   # RectShape's texture scale default (Demo/SceneLoader.cpp:455-461)
            lights.append(struct.pack("<4I4f4f16f", 0, a["shape"], 0, 0, *a["color"], 0.0, *params, *a["transform"]))
        elif kind in ("point_light", "spot_light", "directional_light"):
            k = {"point_light": 1, "spot_light": 2, "directional_light": 3}[kind]
            lights.append(struct.pack("<4I4f4f16f", k, 0, 0, 0, *a["color"], 0.0, a.get("angle", 0.0), 0, 0, 0, *a["transform"]))
        elif kind == "background_light":
            ident = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
            env = texture_index[a["texture"]] + 1 if a.get("texture") is not None else 0   # environment map: BackgroundLight::mTexture
            lights.append(struct.pack("<4I4f4f16f", 4, 0, env, 0, *a["color"], 0.0, 0, 0, 0, 0, *ident))
        else:
            raise ValueError("scene element %r is not exported" % kind)
    c = camera.settings
    with open(path, "wb") as f:
        f.write(b"RTREF002")
        f.write(struct.pack("<14I", width, height, passes, threads, max_ray_depth, min_rr_depth, dimensions, int(use_blue_noise), int(light_sampling_all),
                            len(materials), len(meshes), len(objects), len(lights), int(dump_image)))
        f.write(struct.pack("<2fQ2I", aa_spread, 0.0, seed, len(textures), 0))
        f.write(struct.pack("<3f3f2fI2fI", *c["translation"], *c["orientation_deg"], c["fov_rad"], c["aspect"], int(c["dof"]), c["focal_plane_distance"], c["aperture"], int(c.get("bokeh_shape", 0))))
        for group in (textures, materials, meshes, objects, lights):
            for blob in group:
                f.write(blob)


def run(scene_path, threads=None, passes=None, timeout=900):
    """Runs the binary; returns (stats dict printed by it, dict of the output file: counters, first-pass sampler seeds, image or None)."""
    out_path = scene_path + ".out"
    cmd = [EXE, scene_path, out_path]
    if threads is not None:
        cmd.append(str(int(threads)))
        if passes is not None:
            cmd.append(str(int(passes)))
    # "../Data/BlueNoise128_RGBA16.dat" must resolve from the working directory: a scratch tree <tmp>/Data -> raytracer_amd/data, <tmp>/run
    tree = tempfile.mkdtemp(prefix="rtref_cwd_", dir="/tmp")
    os.symlink(os.path.join(DATA_PARENT, "data"), os.path.join(tree, "Data"))
    cwd = os.path.join(tree, "run")
    os.mkdir(cwd)
    try:
        r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=timeout)
    finally:
        shutil.rmtree(tree, ignore_errors=True)
    if r.returncode != 0:
        raise RuntimeError("ref_render failed (%d): %s" % (r.returncode, r.stderr[-500:]))
    line = r.stdout.strip().splitlines()[-1]
    try:
        stats = json.loads(line)
    except ValueError:
        # a frame with a NaN / inf pixel prints its mean as `nan` / `inf`, which JSON has no word for
        stats = json.loads(line.replace("-nan", "NaN").replace("nan", "NaN").replace("-inf", "-Infinity").replace("inf", "Infinity"))
    raw = np.fromfile(out_path, dtype=np.uint8)
    head = raw[:32].view(np.uint32)
    assert head[0] == 0x54554F52
    w, h, num_seeds, dumped = int(head[1]), int(head[2]), int(head[4]), int(head[5])
    off = 32
    counters = raw[off:off + 32].view(np.uint64); off += 32 + 8
    first_offset = raw[off:off + 8].view(np.float32).copy(); off += 8
    seeds = raw[off:off + 4 * num_seeds].view(np.uint32).copy(); off += 4 * num_seeds
    image = raw[off:off + 12 * w * h].view(np.float32).reshape(h, w, 3).copy() if dumped else None
    os.remove(out_path)
    return stats, dict(numRays=int(counters[0]), numPrimaryRays=int(counters[1]), numShadowRays=int(counters[2]), numShadowRaysHit=int(counters[3]),
                       first_pass_seeds=seeds, first_pass_sample_offset=first_offset, image=image)


def timed_baseline(exe, args, scene, camera, ra, dimensions=64, light_sampling_all=False):
    """bench.py's cpu_baseline: the same workload on every hardware thread and on one, a bounded number of passes each."""
    threads = os.cpu_count() or 1
    with tempfile.TemporaryDirectory(prefix="rtref_", dir="/tmp") as tmp:
        path = os.path.join(tmp, "scene.bin")
        export_scene(path, scene, camera, args.width, args.height, 1, threads, args.depth, dimensions=dimensions, light_sampling_all=light_sampling_all, seed=77, dump_image=False)
        # calibrate with one pass on all threads, then as many passes as fit the budget (at most 24: ~12 s of wall clock on every hardware thread with the default budget)
        s1, _ = run(path, threads, 1)
        passes = int(max(1, min(24, (args.cpu_seconds * 0.6) // max(s1["seconds"], 1e-3))))
        sN, _ = run(path, threads, passes) if passes > 1 else (s1, None)
        # one thread: a quarter-resolution frame of the same scene (a full-size pass would take tens of seconds on one core)
        single = None
        try:
            small = os.path.join(tmp, "scene_small.bin")
            export_scene(small, scene, camera, max(64, args.width // 4), max(36, args.height // 4), 1, 1, args.depth, dimensions=dimensions, light_sampling_all=light_sampling_all,
                         seed=77, dump_image=False)
            single, _ = run(small, 1, 1)
        except Exception:
            single = None
    out = {"value": sN["msamples_per_s"], "unit": "Msamples/s", "cores": sN["threads"], "kind": "reference-partial",
           "sample": "%d full pass(es) of the %dx%d frame on %d threads" % (sN["passes"], args.width, args.height, sN["threads"]),
           "seconds": round(sN["seconds"], 2), "numRays": sN["numRays"],
           "what": "the reference's own AVX2/FMA object code (Viewport::Render, ThreadPool, PathTracerMIS::RenderPixel, traversal, shapes, BSDFs, lights, sampler) "
                   "under a glue translation unit for Scene.cpp / Renderer.cpp (oracle/ref_harness/ref_render.cpp)"}
    if single:
        out["one_thread"] = {"value": single["msamples_per_s"], "unit": "Msamples/s", "sample": "one pass of the %dx%d frame" % (single["width"], single["height"])}
    return out
