"""Writes tests/golden/ref_paths/<name>.bin: every vertex of every path of ONE pass of the reference's own PathTracerMIS::RenderPixel, as its
own path-debugging hook records them (PathDebugData, Core/Rendering/PathDebugging.h:27-53, filled at PathTracerMIS.cpp:377-410 -- the
hook the reference's Demo uses for its path inspector).  oracle/_ref/ref_paths is oracle/ref_harness/ref_render.cpp over a build of the
reference's translation units WITHOUT RT_CONFIGURATION_FINAL (the hook and RenderingContext::pathDebugData exist only there), one
thread.  Build container only.  Regenerate: python tests/golden/make_ref_paths_fixtures.py

File: uint32 magic 'RPV1', width, height, maxRayDepth, samplingAll, dimensions, numVertices, 0; float32 vertices[numVertices][28]:
ray origin xyz, ray direction xyz, hit objectId, subObjectId (bit-cast), distance, u, v, frame position xyz, normal xyz, tangent xyz,
texCoord xy, throughput xyzw, bsdfEvent (bit-cast; unset in a path's last record), 0.  Paths follow each other in the order the pixels
were rendered; a path starts where the ray origin is the camera position."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("RT_FIXTURE_OUT", HERE)   # tests/test_golden_regeneration.py regenerates into a scratch directory and compares
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_render   # noqa: E402
import ref_scenes   # noqa: E402

EXE = ref_render.EXE.replace("ref_render", "ref_paths")
SIZE = (40, 30)
SCENES = ("box_mesh", "cornell", "mesh_2k_all", "mesh_single")


def run_paths(scene_path, w, h):
    tree = tempfile.mkdtemp(prefix="rtref_cwd_", dir="/tmp")
    os.symlink(os.path.join(ref_render.DATA_PARENT, "data"), os.path.join(tree, "Data"))
    cwd = os.path.join(tree, "run")
    os.mkdir(cwd)
    out = scene_path + ".out"
    r = subprocess.run([EXE, scene_path, out, "1", "1"], cwd=cwd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    os.remove(out)
    head = raw[:32].view(np.uint32)
    off = 32 + 32 + 8 + 8 + 4 * int(head[4]) + 12 * w * h
    n = int(raw[off:off + 4].view(np.uint32)[0])
    return raw[off + 4:off + 4 + n * 112].view(np.float32).reshape(n, 28).copy()


if __name__ == "__main__":
    w, h = SIZE
    os.makedirs(os.path.join(OUT, "ref_paths"), exist_ok=True)
    for name in SCENES:
        make, _, _, _, depth, sampling_all, dims = ref_scenes.FIXTURES[name]
        scene, camera = make(w / h)
        path = "/tmp/ref_paths_%s.bin" % name
        ref_render.export_scene(path, scene, camera, w, h, 1, 1, depth, dimensions=dims, light_sampling_all=sampling_all, seed=ref_scenes.SEED)
        v = run_paths(path, w, h)
        os.remove(path)
        with open(os.path.join(OUT, "ref_paths", name + ".bin"), "wb") as f:
            f.write(struct.pack("<8I", 0x31565052, w, h, depth, int(sampling_all), dims, len(v), 0))
            f.write(v.astype("<f4").tobytes())
        print(name, len(v), "vertices")
