"""Replays tools/oracle_fuzz.py's case stream for a seed WITHOUT rendering (the stream depends on the generator alone), and renders only the cases whose index is given --
against the oracle, under the environment as it is.  For chasing a soak mismatch:
   python tools/oracle_fuzz_replay.py <seed> list [max cases]              prints index + parameters of every case
   python tools/oracle_fuzz_replay.py <seed> <index> [<index> ...]         renders those cases; prints differing words, first differing pixels, counters
   (FUZZ_KINDS=6 for seeds run by tools/oracle_fuzz.py since the end of round 6: its stream draws from six scene kinds, before that from five)
Test infrastructure (uses the oracle)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import scenes
import oracle_lib, scene_zoo


def stream(seed, limit=None, kinds=5):
    """The case stream of a seed.  kinds = 5: the stream of rounds 1-6 (old seeds replay as they ran); kinds = 6 adds the all-lights x all-BSDFs scene of analytic
    shapes (tests/scene_zoo.py all_lights_scene); kinds = 7: that, and random cameras for the two-level scenes too -- tools/oracle_fuzz.py's default since the end of
    round 6 (a mismatch line names seed, index and kinds); kinds = 8 adds tests/scene_zoo.py random_scene (random materials, shapes, mesh patches, lights, all under general
    rotations) as a seventh scene kind."""
    rng = np.random.RandomState(seed)
    index = -1
    while limit is None or index + 1 < limit:
        index += 1
        kind = rng.randint(7 if kinds >= 8 else min(kinds, 6))
        w, h = [(64, 48), (128, 72), (160, 96), (96, 160)][rng.randint(4)]
        if kind == 0: make = ("sponza", int(rng.choice([300, 3000, 20000])), int(rng.randint(1, 1000)), False, False)
        elif kind == 4: make = ("sponza", int(rng.choice([300, 3000])), int(rng.randint(1, 1000)), True, bool(rng.randint(2)))
        elif kind == 1: make = ("mesh_scene", int(rng.choice([2000, 8000])))
        elif kind == 2: make = ("cornell",)
        elif kind == 5: make = ("zoo",)
        elif kind == 6: make = ("random", int(rng.randint(1, 1 << 30)))
        else: make = ("sphere",)
        cam = None
        if kind in (0, 4):
            cam = ((float(rng.uniform(-13, 13)), float(rng.uniform(0.3, 10)), float(rng.uniform(-5, 5))), (float(rng.uniform(-60, 60)), float(rng.uniform(0, 360)), 0.0), float(rng.uniform(30, 100)))
        elif kinds >= 7 and kind in (1, 2, 5):
            # kinds = 7 (end of round 6): the two-level scenes seen from a camera somewhere around their own one too (the mismatch of seed 9606 was a geometric
            # coincidence of ONE fixed camera: more viewpoints, more coincidences)
            base = {1: ((-12.5, 2.2, 0.6), (4.0, 82.0, 0.0), 65.0, (2.0, 1.2, 2.0)), 2: ((0.0, 0.0, 6.0), (0.0, 180.0, 0.0), 40.0, (0.6, 0.6, 1.5)), 5: ((0.5, 2.5, 9.0), (12.0, 180.0, 0.0), 55.0, (3.0, 1.5, 2.5))}[kind]
            jitter = [float(rng.uniform(-1.0, 1.0)) for _ in range(6)]
            if rng.randint(4) != 0:
                cam = (tuple(base[0][a] + jitter[a] * base[3][a] for a in range(3)), (base[1][0] + 15.0 * jitter[3], base[1][1] + 30.0 * jitter[4], 0.0), base[2] * (1.0 + 0.4 * jitter[5]))
        args = dict(max_ray_depth=int(rng.choice([0, 2, 6, 10])), min_russian_roulette_depth=int(rng.choice([1, 4, 20])), light_sampling_all=bool(rng.randint(2)),
                    dimensions=int(rng.choice([16, 64, 128])), use_blue_noise=bool(rng.randint(2)))
        passes = int(rng.choice([1, 2, 4]))
        counters_on = bool(rng.randint(2))
        vp_seed = int(rng.randint(1, 1 << 30))
        schedule = (int(rng.choice([-1, -1, 0, 1, 2, 3, 4])), int(rng.choice([-1, 0, 1])))
        yield index, kind, w, h, make, cam, args, passes, counters_on, vp_seed, schedule


def build(case):
    index, kind, w, h, make, cam, args, passes, counters_on, vp_seed, schedule = case
    if make[0] == "sponza": scene, camera = scenes.sponza_class(w / h, make[1], seed=make[2], textured=make[3], extra_texture=make[4])
    elif make[0] == "mesh_scene": scene, camera = scene_zoo.mesh_scene(w / h, triangles=make[1])
    elif make[0] == "cornell": scene, camera = scenes.cornell_box(w / h)
    elif make[0] == "zoo": scene, camera = scene_zoo.all_lights_scene(w / h)
    elif make[0] == "random": scene, camera = scene_zoo.random_scene(w / h, make[1])
    else: scene, camera = scenes.sphere_area_light(w / h)
    if cam: camera = ra.Camera(cam[0], cam[1], w / h, cam[2])
    return scene, camera


def render(case, quiet=False):
    index, kind, w, h, make, cam, args, passes, counters_on, vp_seed, schedule = case
    scene, camera = build(case)
    bn = ra.load_blue_noise()
    desc = scene.desc; desc.contents.blueNoise = bn.ctypes.data
    vp = ra.Viewport(w, h, seed=vp_seed, **args)
    vp.set_renderer(scene, intersection_counters=counters_on)
    ra.rtgpu_lib().rtgpu_set_schedule(vp.device_context(), C.c_uint32(0), C.c_int32(schedule[0]))
    ra.rtgpu_lib().rtgpu_set_schedule(vp.device_context(), C.c_uint32(1), C.c_int32(schedule[1]))
    ref = np.zeros((h, w, 3), dtype=np.float32); ref2 = np.zeros((h, w, 3), dtype=np.float32); cnt = np.zeros(16, dtype=np.uint64)
    for _ in range(passes):
        p = vp.next_pass_params(camera)
        vp.render_pass_with(p)
        oracle_lib.render_pass(desc, p, w, h, ref, ref2, cnt, threads=min(64, os.cpu_count() or 1))
    img, img2 = vp.sum_buffer(secondary=True)
    c = vp.counters()
    diff = np.argwhere(img.view(np.uint32) != ref.view(np.uint32))
    names = ra.COUNTER_NAMES[:4] if not counters_on else ra.COUNTER_NAMES[:12]
    same = len(diff) == 0 and np.array_equal(img2.view(np.uint32), ref2.view(np.uint32)) and all(c[n] == int(cnt[i]) for i, n in enumerate(ra.COUNTER_NAMES[:12]) if n in names)
    if quiet:
        return same, len(diff)
    print("case %d: %d differing words; counters gpu %s oracle %s" % (index, len(diff), {n: c[n] for n in ra.COUNTER_NAMES[:4]}, {n: int(cnt[i]) for i, n in enumerate(ra.COUNTER_NAMES[:4])}),
          "retraced", c.get("numRetracedRays"), flush=True)
    for y, x, ch in diff[:6]:
        print("   pixel (%d, %d) channel %d: gpu %.9g oracle %.9g" % (x, y, ch, img[y, x, ch], ref[y, x, ch]))
    return same, len(diff)


if __name__ == "__main__":
    seed = int(sys.argv[1])
    kinds = int(os.environ.get("FUZZ_KINDS", "5"))     # 6 / 7: the streams tools/oracle_fuzz.py drew at the end of round 6 (it prints the value to use with every mismatch)
    if sys.argv[2] == "list":
        for case in stream(seed, int(sys.argv[3]) if len(sys.argv) > 3 else 1000, kinds):
            print(case)
    else:
        wanted = set(int(a) for a in sys.argv[2:])
        for case in stream(seed, max(wanted) + 1, kinds):
            if case[0] in wanted:
                print(case, "env", {k: v for k, v in os.environ.items() if k.startswith("RTGPU_")})
                render(case)
