"""Soak test: the device (library defaults: dense path state, the 4-wide walks, four lanes, intersection counters off in half of the cases)
against the CPU oracle on random scenes, cameras, sizes and settings -- bit-identical sum buffers and ray counters expected.
usage: python tools/oracle_fuzz.py [seconds] [seed]      (tests/test_gpu_fuzz.py runs a bounded slice of it under `pytest -m gpu`)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import scenes
import oracle_lib, scene_zoo


def run(budget=300.0, seed=1, min_cases=0, log=print):
    """Random cases until `budget` seconds are used up (and at least `min_cases`); returns (cases, mismatches, default-walk cases)."""
    rng = np.random.RandomState(seed)
    t_end = time.time() + budget
    cases = bad = default_walk = 0
    bn = ra.load_blue_noise()
    threads = min(64, os.cpu_count() or 1)
    while time.time() < t_end or cases < min_cases:
        kind = rng.randint(5)
        w, h = [(64, 48), (128, 72), (160, 96), (96, 160)][rng.randint(4)]
        if kind == 0: scene, camera = scenes.sponza_class(w / h, int(rng.choice([300, 3000, 20000])), seed=int(rng.randint(1, 1000)))
        elif kind == 4:   # textured: the "lean + simple bitmaps" / "lean + textures" shading kernels
            scene, camera = scenes.sponza_class(w / h, int(rng.choice([300, 3000])), seed=int(rng.randint(1, 1000)), textured=True, extra_texture=bool(rng.randint(2)))
        elif kind == 1: scene, camera = scene_zoo.mesh_scene(w / h, triangles=int(rng.choice([2000, 8000])))
        elif kind == 2: scene, camera = scenes.cornell_box(w / h)
        else: scene, camera = scenes.sphere_area_light(w / h)
        if kind in (0, 4):
            camera = ra.Camera((float(rng.uniform(-13, 13)), float(rng.uniform(0.3, 10)), float(rng.uniform(-5, 5))), (float(rng.uniform(-60, 60)), float(rng.uniform(0, 360)), 0.0),
                               w / h, float(rng.uniform(30, 100)))
        args = dict(max_ray_depth=int(rng.choice([0, 2, 6, 10])), min_russian_roulette_depth=int(rng.choice([1, 4, 20])), light_sampling_all=bool(rng.randint(2)),
                    dimensions=int(rng.choice([16, 64, 128])), use_blue_noise=bool(rng.randint(2)))
        passes = int(rng.choice([1, 2, 4]))
        counters_on = bool(rng.randint(2))
        desc = scene.desc
        desc.contents.blueNoise = bn.ctypes.data
        vp = ra.Viewport(w, h, seed=int(rng.randint(1, 1 << 30)), **args)
        vp.set_renderer(scene, intersection_counters=counters_on)
        # round 4's launch-sequence variants: the fused tail taking over at a random bounce (or by policy, or never), the block-local re-trace on / off / by policy
        import ctypes as C
        schedule = (int(rng.choice([-1, -1, 0, 1, 2, 3, 4])), int(rng.choice([-1, 0, 1])))
        ra.rtgpu_lib().rtgpu_set_schedule(vp.device_context(), C.c_uint32(0), C.c_int32(schedule[0]))
        ra.rtgpu_lib().rtgpu_set_schedule(vp.device_context(), C.c_uint32(1), C.c_int32(schedule[1]))
        ref = np.zeros((h, w, 3), dtype=np.float32); ref2 = np.zeros((h, w, 3), dtype=np.float32)
        cnt = np.zeros(16, dtype=np.uint64)
        for _ in range(passes):
            p = vp.next_pass_params(camera)
            vp.render_pass_with(p)
            oracle_lib.render_pass(desc, p, w, h, ref, ref2, cnt, threads=threads)
        img, img2 = vp.sum_buffer(secondary=True)
        c = vp.counters()
        names = ra.COUNTER_NAMES[:4] if not counters_on else [n for n in ra.COUNTER_NAMES[:12]]
        same = np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and np.array_equal(img2.view(np.uint32), ref2.view(np.uint32)) and \
            all(c[n] == int(cnt[i]) for i, n in enumerate(ra.COUNTER_NAMES[:12]) if n in names)
        cases += 1
        default_walk += 0 if counters_on else 1
        if not same:
            bad += 1
            log("MISMATCH", kind, w, h, args, passes, counters_on, schedule, int(np.count_nonzero(img.view(np.uint32) != ref.view(np.uint32))))
    return cases, bad, default_walk


if __name__ == "__main__":
    cases, bad, default_walk = run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("cases %d (%d with the default walk), mismatches %d" % (cases, default_walk, bad))
    sys.exit(1 if bad else 0)
