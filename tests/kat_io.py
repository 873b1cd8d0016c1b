import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_kat(name):
    raw = np.fromfile(os.path.join(GOLDEN, name), dtype=np.uint32)
    magic, func, n, ins, outs, _ = (int(v) for v in raw[:6])
    assert magic == 0x3154414B, "bad KAT magic in %s" % name
    data = raw[6:].view(np.float32)
    return func, data[:n * ins].reshape(n, ins).copy(), data[n * ins:n * ins + n * outs].reshape(n, outs).copy()


def bit_mismatch(expected, got):
    """Boolean mask of values that differ bitwise (NaN == NaN counts as equal)."""
    e, g = np.ascontiguousarray(expected, dtype=np.float32), np.ascontiguousarray(got, dtype=np.float32)
    return ~((e.view(np.uint32) == g.view(np.uint32)) | (np.isnan(e) & np.isnan(g)))


def load_texture_kat():
    """tests/golden/texture_kat.bin (layout: oracle/ref_harness/kat_gen.cpp genTextures).  Returns a dict with the
    RtTexture array, the texel blob (uint8) and the three record tables as structured numpy arrays."""
    import ctypes as C
    import raytracer_amd as ra
    raw = np.fromfile(os.path.join(GOLDEN, "texture_kat.bin"), dtype=np.uint8)
    magic, num_tex, num_eval, num_mat, num_bg, _ = (int(v) for v in raw[:24].view(np.uint32))
    assert magic == 0x31584554
    texel_bytes = int(raw[24:32].view(np.uint64)[0])
    off = 32
    tex_size = C.sizeof(ra.RtTexture)
    textures = (ra.RtTexture * num_tex).from_buffer_copy(raw[off:off + num_tex * tex_size].tobytes())
    off += num_tex * tex_size
    texels = raw[off:off + texel_bytes].copy()
    off += texel_bytes
    ev = np.dtype([("texture", np.uint32), ("uv", np.float32, 2), ("out", np.float32, 4)])
    evals = raw[off:off + num_eval * ev.itemsize].view(ev)
    off += num_eval * ev.itemsize
    mt = np.dtype([("material", np.uint8, C.sizeof(ra.RtMaterial)), ("uv", np.float32, 2), ("out", np.float32, 14)])
    mats = raw[off:off + num_mat * mt.itemsize].view(mt)
    off += num_mat * mt.itemsize
    bg = np.dtype([("light", np.uint8, C.sizeof(ra.RtLight)), ("dir", np.float32, 4), ("out", np.float32, 4)])
    bgs = raw[off:off + num_bg * bg.itemsize].view(bg)
    off += num_bg * bg.itemsize
    assert off == raw.size
    return dict(textures=textures, texels=texels, evals=evals, materials=mats, backgrounds=bgs)


def mesh_fixture_scene():
    """The mesh the reference's MeshShape::Initialize was fed for mesh_kat.bin (mesh_input.bin), as a one-object scene."""
    import raytracer_amd as ra
    raw = open(os.path.join(GOLDEN, "mesh_input.bin"), "rb").read()
    nv, nt, nmat = (int(v) for v in np.frombuffer(raw[:12], dtype=np.uint32))
    off = 12

    def take(count, dtype, width):
        nonlocal off
        a = np.frombuffer(raw[off:off + count * width * 4], dtype=dtype).reshape(count, width).copy()
        off += count * width * 4
        return a
    pos, nrm, tan, uv = take(nv, np.float32, 3), take(nv, np.float32, 3), take(nv, np.float32, 3), take(nv, np.float32, 2)
    idx, mat = take(nt, np.uint32, 3), take(nt, np.uint32, 1).reshape(-1)
    scene = ra.Scene()
    mats = [scene.add_material("diffuse") for _ in range(nmat)]
    scene.add_mesh(pos, idx, nrm, tan, uv, mat, mats)
    scene.build()
    return scene, mats, nt
