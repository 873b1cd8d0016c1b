"""Writes tests/golden/mesh_input.bin: the mesh that oracle/ref_harness/kat_gen.cpp feeds to the reference's
MeshShape::Initialize for the mesh-path KAT.  Run before `make -C oracle/ref_harness fixtures`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raytracer_amd import scenes  # noqa: E402

pos, idx, nrm, tan, uv, mat = scenes.sponza_class_mesh(target_triangles=6000, seed=3)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mesh_input.bin")
with open(out, "wb") as f:
    np.array([pos.shape[0], idx.shape[0], len(scenes.SPONZA_MATERIALS)], dtype=np.uint32).tofile(f)
    pos.astype(np.float32).tofile(f); nrm.astype(np.float32).tofile(f); tan.astype(np.float32).tofile(f); uv.astype(np.float32).tofile(f)
    idx.astype(np.uint32).tofile(f); mat.astype(np.uint32).tofile(f)
print("wrote", out, pos.shape[0], "vertices", idx.shape[0], "triangles")
