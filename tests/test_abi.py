"""CPU-side checks of the boundary: the C-ABI library loads without a GPU, exports every symbol declared in
include/rtgpu.h, struct layouts agree between C, the oracle and the ctypes mirrors, and the product fails
loudly (no fallback) when no device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rtgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtgpu_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    import raytracer_amd as ra
    lib = ra.rtgpu_lib()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "librtgpu.so does not export %s" % n
    assert lib.rtgpu_abi_version() == 3


def test_struct_layouts_agree(built):
    import raytracer_amd as ra
    import oracle_lib
    o = oracle_lib.lib()
    for what, ty in enumerate((ra.RtSceneDesc, ra.RtPassParams, ra.RtObject, ra.RtLight, ra.RtMaterial, ra.RtCamera, ra.RtTexture)):
        assert o.rto_sizeof(what) == C.sizeof(ty), ty.__name__
    assert C.sizeof(ra.RtNode) == 32 and C.sizeof(ra.RtMaterial) == 80 and C.sizeof(ra.RtTexture) == 96 and C.sizeof(ra.RtCounters) == 128


def test_no_gpu_means_loud_failure_not_fallback(built):
    import raytracer_amd as ra
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = ra.rtgpu_lib()
    ctx = C.c_void_p()
    assert lib.rtgpu_create(0, C.byref(ctx)) == -2        # RTGPU_ERR_NO_DEVICE
    assert b"no HIP device" in lib.rtgpu_last_error()
    from raytracer_amd import scenes
    scene, camera = scenes.sphere_area_light(1.0)
    vp = ra.Viewport(16, 16, seed=1)
    with pytest.raises(RuntimeError):
        vp.set_renderer(scene)


def test_product_never_touches_the_oracle():
    """Nothing under raytracer_amd/ or include/ may reference oracle/ (the oracle is test infrastructure)."""
    offenders = []
    for base in ("raytracer_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".inl")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"oracle/|oracle_lib|liboracle|rto_", text):
                        offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_scene_flattening_matches_reference_layout(built):
    """Cornell box: 10 traceable objects (9 shapes + the finite rect light) in BVH leaf order, one light, no global lights."""
    import raytracer_amd as ra
    from raytracer_amd import scenes
    scene, _ = scenes.cornell_box(4.0 / 3.0)
    d = scene.desc.contents
    assert (d.numObjects, d.numLights, d.numGlobalLights, d.numMeshes) == (10, 1, 0, 0)
    kinds = sorted((d.objects[i].objectKind, d.objects[i].shapeKind) for i in range(10))
    assert kinds.count((1, 2)) == 1 and kinds.count((0, 0)) == 2 and kinds.count((0, 1)) == 7
    # every leaf of the top-level BVH references a valid object range and each object exactly once
    seen = []
    for i in range(d.numTopNodes):
        n = d.topNodes[i]
        leaves = n.leaves & 0x3FFFFFFF
        if leaves and i != 1:
            seen += list(range(n.childIndex, n.childIndex + leaves))
    assert sorted(seen) == list(range(10))
    # inverse transforms really are inverses
    for i in range(10):
        m = np.array(d.objects[i].transform[:]).reshape(4, 4)
        inv = np.array(d.objects[i].invTransform[:]).reshape(4, 4)
        assert np.allclose(m @ inv, np.eye(4), atol=1e-5)
