"""Soak test of the 4-wide walk's exactness: random meshes, cameras, sizes, depths and sampling settings rendered twice by the device --
RTGPU_WIDE=1 (k_trace_wide + re-trace) and RTGPU_WIDE=0 (the binary walk in the reference's order) -- must give bit-identical sum buffers
and ray counters.  usage: python tools/wide_fuzz.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import scenes

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = bad = 0
total_rays = total_retraced = 0
while time.time() < t_end:
    tris = int(rng.choice([300, 2000, 8000, 30000, 120000, 262144]))
    w, h = [(64, 48), (200, 120), (320, 200), (640, 360), (960, 540)][rng.randint(5)]
    depth = int(rng.choice([1, 3, 8, 12]))
    all_lights = bool(rng.randint(2))
    rr = int(rng.choice([1, 3, 20]))
    passes = int(rng.choice([1, 2, 5]))
    mesh_seed = int(rng.randint(1, 1000))
    scene, camera = scenes.sponza_class(w / h, tris, seed=mesh_seed)
    # a camera somewhere inside the atrium, looking anywhere
    pos = (float(rng.uniform(-13, 13)), float(rng.uniform(0.3, 10)), float(rng.uniform(-5, 5)))
    rot = (float(rng.uniform(-60, 60)), float(rng.uniform(0, 360)), 0.0)
    camera = ra.Camera(pos, rot, w / h, float(rng.uniform(30, 100)))
    seed = int(rng.randint(1, 1 << 30))
    out = []
    for wide in ("1", "0"):
        os.environ["RTGPU_WIDE"] = wide
        vp = ra.Viewport(w, h, seed=seed, max_ray_depth=depth, min_russian_roulette_depth=rr, light_sampling_all=all_lights, dimensions=128)
        vp.set_renderer(scene, intersection_counters=False)
        vp.render(camera, passes)
        out.append((vp.sum_buffer(secondary=True), vp.counters()))
    (a, a2), ca = out[0]; (b, b2), cb = out[1]
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a2.view(np.uint32), b2.view(np.uint32)) and \
        all(ca[k] == cb[k] for k in ("numRays", "numShadowRays", "numShadowRaysHit", "numPrimaryRays", "numMeshHits"))
    cases += 1
    total_rays += ca["numRays"] + ca["numShadowRays"]; total_retraced += ca["numRetracedRays"]
    if not same or cb["numRetracedRays"] != 0:
        bad += 1
        print("MISMATCH", dict(tris=tris, w=w, h=h, depth=depth, all_lights=all_lights, rr=rr, passes=passes, mesh_seed=mesh_seed, pos=pos, rot=rot, seed=seed),
              "differing values", int(np.count_nonzero(a.view(np.uint32) != b.view(np.uint32))), flush=True)
print("cases %d, mismatches %d, rays %d, re-traced %d (%.3f %%)" % (cases, bad, total_rays, total_retraced, 100.0 * total_retraced / max(1, total_rays)))
sys.exit(1 if bad else 0)
