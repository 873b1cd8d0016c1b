"""Writes tests/golden/ref_render/<name>.bin: what the REFERENCE'S OWN renderer (Viewport::Render -> PathTracerMIS::RenderPixel, the
objects of oracle/_ref/libref_partial.a behind oracle/_ref/ref_render; see oracle/ref_harness/ref_render.cpp for the glue) makes of the
scenes of tests/ref_scenes.py: the float3 sum buffer, the ray counters, the sampler seeds and the anti-aliasing offset of the first
pass.  Build container only (needs oracle/_ref/ref_render).  Regenerate: python tests/golden/make_ref_render_fixtures.py

File: uint32 magic 'RRF1', width, height, passes, maxRayDepth, samplingAll, dimensions, numSeeds; uint64 numRays, numPrimaryRays,
numShadowRays, numShadowRaysHit; float32 sampleOffset[2]; uint32 seeds[numSeeds]; float32 image[height][width][3]."""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("RT_FIXTURE_OUT", HERE)   # tests/test_golden_regeneration.py regenerates into a scratch directory and compares
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_render   # noqa: E402
import ref_scenes   # noqa: E402

def write_fixture(name, w, h, passes, depth, sampling_all, dims, out):
    with open(os.path.join(OUT, "ref_render", name + ".bin"), "wb") as f:
        f.write(struct.pack("<8I", 0x31465252, w, h, passes, depth, int(sampling_all), dims, len(out["first_pass_seeds"])))
        f.write(struct.pack("<4Q", out["numRays"], out["numPrimaryRays"], out["numShadowRays"], out["numShadowRaysHit"]))
        f.write(out["first_pass_sample_offset"].astype("<f4").tobytes())
        f.write(out["first_pass_seeds"].astype("<u4").tobytes())
        f.write(out["image"].astype("<f4").tobytes())


def make_statistical(names=None):
    for name, (make, w, h, passes, depth, sampling_all, dims) in ref_scenes.STATISTICAL_FIXTURES.items():
        if names and name not in names:
            continue
        scene, camera = make(w / h)
        path = "/tmp/ref_fixture_%s.bin" % name
        ref_render.export_scene(path, scene, camera, w, h, passes, 1, depth, dimensions=dims, light_sampling_all=sampling_all, seed=ref_scenes.SEED)
        stats, out = ref_render.run(path, threads=1)       # one thread: the per-thread generator's light picks are then a function of the seed
        os.remove(path)
        write_fixture(name, w, h, passes, depth, sampling_all, dims, out)
        print(name, stats["mean"], out["numRays"], out["numShadowRays"])


if __name__ == "__main__":
    # no argument: everything; "statistical": only the entropy-seeded fixture (NOT reproducible byte for byte: the reference's per-thread generator is seeded
    # from std::random_device); "exact": only the fixtures that are a function of the seed (what tests/test_golden_regeneration.py byte-compares)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "statistical"):
        make_statistical()
    if which == "statistical":
        sys.exit(0)
    for name, (make, w, h, passes, depth, sampling_all, dims) in ref_scenes.FIXTURES.items():
        scene, camera = make(w / h)
        path = "/tmp/ref_fixture_%s.bin" % name
        ref_render.export_scene(path, scene, camera, w, h, passes, 4, depth, dimensions=dims, light_sampling_all=sampling_all, seed=ref_scenes.SEED)
        stats, out = ref_render.run(path)
        again, out2 = ref_render.run(path, threads=1)      # the result must not depend on the thread count
        assert np.array_equal(out["image"], out2["image"]), name
        os.remove(path)
        write_fixture(name, w, h, passes, depth, sampling_all, dims, out)
        print(name, stats["mean"], out["numRays"], out["numShadowRays"])
