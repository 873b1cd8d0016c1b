"""Soak of the ORACLE against the REFERENCE'S OWN RENDERER (CPU only, build container only: needs oracle/_ref/ref_render, i.e. /root/reference): the case stream of
tools/oracle_fuzz_replay.py (random scenes, cameras, sizes, depths, roulette settings), every case rendered by oracle/_ref/ref_render -- the reference's compiled
Viewport::Render / PathTracerMIS / traversal / shading objects, one thread -- and by the oracle in its x86 approximation mode (the reference's _mm_rcp_ss / _mm_rsqrt_ps
through the host's instructions).  Expected: every pixel and the ray counters bit for bit.  LightSamplingStrategy::All is forced (under `Single` with several lights the
reference picks the light with an entropy-seeded per-thread generator: not a function of the seed).
   python tools/reference_fuzz.py [seconds] [seed] [kinds] [only this scene kind, e.g. random]
Test infrastructure; nothing here is on the product path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import raytracer_amd as ra
import oracle_lib
import ref_render
import oracle_fuzz_replay as replay


def run(budget=120.0, seed=1, kinds=8, log=print, only=None, min_cases=0):
    ok, signature = oracle_lib.set_x86_approximations(True)
    assert ok, "this CPU cannot evaluate the reference's approximate instructions"
    bn = ra.load_blue_noise()
    t_end = time.time() + budget
    cases = bad = 0
    try:
        for case in replay.stream(seed, None, kinds):
            if time.time() >= t_end and cases >= min_cases:
                break
            index, kind, w, h, make, cam, args, passes, counters_on, vp_seed, schedule = case
            if only and make[0] != only:
                continue
            if make[0] == "sponza" and make[3] and not os.environ.get("REFERENCE_FUZZ_TEXTURED_SPONZA"):
                continue     # (textured atrium variants: a second of exporting per case; REFERENCE_FUZZ_TEXTURED_SPONZA=1 includes them -- random scenes carry bitmaps anyway)
            scene, camera = replay.build(case)
            # lens settings are this tool's own draw (the replay stream stays what the recorded seeds mean): a third of the cases with depth of field, all three
            # bokeh shapes.  (Barrel distortion with a variable factor draws from the reference's entropy-seeded generator: not comparable.)
            lens = np.random.default_rng([seed, index, 77])
            if lens.random() < 0.34:
                camera.set_lens(int(lens.integers(0, 3)))
                camera.set_dof(True, float(np.float32(lens.uniform(1.0, 12.0))), float(np.float32(lens.uniform(0.01, 0.3))))
            single_light = scene.desc.contents.numLights <= 1
            sampling_all = True if not single_light else args["light_sampling_all"]
            # the reference draws from an entropy-seeded per-thread generator once a path has used up its sampler dimensions (GenericSampler::GetInt's fallback):
            # the comparison needs paths that stay inside them -- 128 dimensions, the depth cut to what they hold (4 for the camera, per vertex 3 per sampled light +
            # 1 for the roulette + 3 for the BSDF)
            dims = 128
            per_vertex = 3 * (scene.desc.contents.numLights if sampling_all else 1) + 4
            deepest = (dims - 4) // per_vertex - 1
            if deepest < 0:
                continue
            args = dict(args, max_ray_depth=min(args["max_ray_depth"], deepest))
            path = "/tmp/reference_fuzz_%d.bin" % os.getpid()
            try:
                ref_render.export_scene(path, scene, camera, w, h, passes, 1, args["max_ray_depth"], min_rr_depth=args["min_russian_roulette_depth"], dimensions=dims,
                                        use_blue_noise=args["use_blue_noise"], light_sampling_all=sampling_all, seed=vp_seed)
            except (KeyError, ValueError):
                continue     # a texture kind the exporter does not write (checkerboard / noise / mix, filters other than the default: the harness builds plain bitmaps only)
            stats, out = ref_render.run(path, threads=1)
            os.remove(path)
            scene.desc.contents.blueNoise = bn.ctypes.data
            vp = ra.Viewport(w, h, seed=vp_seed, max_ray_depth=args["max_ray_depth"], min_russian_roulette_depth=args["min_russian_roulette_depth"], light_sampling_all=sampling_all,
                             dimensions=dims, use_blue_noise=args["use_blue_noise"])
            vp.reset()
            img = np.zeros((h, w, 3), dtype=np.float32); cnt = np.zeros(16, dtype=np.uint64)
            seeds = None
            for _ in range(passes):     # the stream's 1 / 2 / 4 passes: the sum buffer after all of them (the Halton sequence's progression is the host mirror's)
                p = vp.next_pass_params(camera)
                if seeds is None:
                    seeds = np.ctypeslib.as_array(p.seed, shape=(p.numDimensions,)).copy()
                oracle_lib.render_pass(scene.desc, p, w, h, img, None, cnt, threads=min(32, os.cpu_count() or 1))
            differing = int(np.count_nonzero((img.view(np.uint32) != out["image"].view(np.uint32)).any(axis=2)))
            same = differing == 0 and np.array_equal(seeds, out["first_pass_seeds"]) and int(cnt[0]) == out["numRays"] and int(cnt[1]) == out["numShadowRays"] and int(cnt[2]) == out["numShadowRaysHit"]
            cases += 1
            if not same:
                bad += 1
                log("MISMATCH seed %d index %d kinds %d" % (seed, index, kinds), make, (w, h), args, "pixels differing", differing, "numRays", int(cnt[0]), out["numRays"],
                    "shadow", int(cnt[1]), out["numShadowRays"], int(cnt[2]), out["numShadowRaysHit"])
    finally:
        oracle_lib.set_x86_approximations(False)
    return cases, bad, signature


if __name__ == "__main__":
    cases, bad, signature = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(sys.argv[3]) if len(sys.argv) > 3 else 8, only=sys.argv[4] if len(sys.argv) > 4 else None)
    print("cases %d, mismatches %d (approximation tables %08x %08x)" % (cases, bad, signature[0], signature[1]))
    sys.exit(1 if bad else 0)
