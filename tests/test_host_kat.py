"""Pins the host-side algorithms of the product (BVH builder, Halton sequence, Random, transforms, mesh
preprocessing) and the oracle's mesh traversal against golden vectors produced by the reference's own code
(tests/golden/, see README.md there).  CPU only: loads the host library, never creates a device context."""
import ctypes as C
import os

import numpy as np
import pytest

import kat_io
import oracle_lib


@pytest.fixture(scope="module")
def host(built):
    import raytracer_amd as ra
    return ra.host_lib()


def test_random_matches_reference(host):
    raw = np.fromfile(os.path.join(kat_io.GOLDEN, "random.bin"), dtype=np.uint8)
    scalar = raw[:16].view(np.uint64).copy()
    simd = raw[16:48].view(np.uint64).copy()
    count = int(raw[48:52].view(np.uint32)[0])
    off = 52
    longs = raw[off:off + 8 * count].view(np.uint64); off += 8 * count
    vec4 = raw[off:off + 16 * count].view(np.float32); off += 16 * count
    out_l = np.zeros(count, dtype=np.uint64)
    out_v = np.zeros(4 * count, dtype=np.float32)
    host.rth_kat_random(scalar.ctypes.data_as(C.POINTER(C.c_uint64)), simd.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(count),
                        out_l.ctypes.data_as(C.POINTER(C.c_uint64)), out_v.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(out_l, longs)
    assert np.array_equal(out_v.view(np.uint32), vec4.view(np.uint32))


@pytest.mark.parametrize("dims", [64, 128])
def test_halton_seeds_bit_exact(host, dims):
    raw = np.fromfile(os.path.join(kat_io.GOLDEN, "halton_%d.bin" % dims), dtype=np.uint32)
    d, passes = int(raw[0]), int(raw[1])
    assert d == dims
    state = raw[2:6].view(np.uint64).copy()
    expected = raw[6:6 + d * passes]
    out = np.zeros(d * passes, dtype=np.uint32)
    host.rth_kat_halton(state.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(d), C.c_uint32(passes), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert np.array_equal(out, expected)


def test_float_normal2_bit_exact(host):
    func, inputs, expected = kat_io.load_kat("math_float_normal2.kat")
    for i in range(inputs.shape[0]):
        out = (C.c_float * 4)()
        host.rth_kat_float_normal2(C.c_float(inputs[i, 0]), C.c_float(inputs[i, 1]), out)
        assert not kat_io.bit_mismatch(expected[i], np.array(out[:], dtype=np.float32)).any()


def test_transforms_match_the_reference_bit_for_bit(host):
    # host transforms are INPUTS of the C-ABI (RtObject::transform / invTransform): a render only equals the reference's bit for bit if
    # they do.  Until round 6 the inverse was "a few ulp" (a differently rounded formula) -- found by tools/reference_fuzz.py on rotated box lights.
    _, inputs, expected = kat_io.load_kat("host_euler.kat")
    for i in range(inputs.shape[0]):
        out = (C.c_float * 16)()
        host.rth_transform_from_euler((C.c_float * 3)(*inputs[i, :3]), (C.c_float * 3)(*inputs[i, 3:6]), out)
        assert np.array_equal(np.array(out[:], dtype=np.float32).view(np.uint32), expected[i].astype(np.float32).view(np.uint32)), i
    _, inputs, expected = kat_io.load_kat("host_inverse.kat")
    for i in range(inputs.shape[0]):
        out = (C.c_float * 16)()
        host.rth_matrix_inverse((C.c_float * 16)(*inputs[i]), out)
        assert np.array_equal(np.array(out[:], dtype=np.float32).view(np.uint32), expected[i].astype(np.float32).view(np.uint32)), i


def _iter_bvh_cases():
    raw = np.fromfile(os.path.join(kat_io.GOLDEN, "bvh_builder.bin"), dtype=np.uint32)
    num = int(raw[0]); off = 1
    for _ in range(num):
        n, num_nodes = int(raw[off]), int(raw[off + 1]); off += 2
        boxes = raw[off:off + 6 * n].view(np.float32).reshape(n, 6).copy(); off += 6 * n
        nodes = raw[off:off + 8 * num_nodes].reshape(num_nodes, 8).copy(); off += 8 * num_nodes
        order = raw[off:off + n].copy(); off += n
        yield n, boxes, nodes, order


def _mask_unused(nodes):
    """Node 1 is never written (root at 0, child pairs from 2) and a leaf's splitAxis is uninitialised in the reference."""
    nodes = nodes.copy()
    if len(nodes) > 1:
        nodes[1] = 0
    leaf = (nodes[:, 7] & 0x3FFFFFFF) != 0
    nodes[leaf, 7] &= 0x3FFFFFFF
    return nodes


def test_bvh_builder_matches_reference(host):
    for n, boxes, ref_nodes, ref_order in _iter_bvh_cases():
        out_nodes = np.zeros((max(2 * n, 2), 8), dtype=np.uint32)
        out_order = np.zeros(n, dtype=np.uint32)
        num_nodes = C.c_uint32(0)
        r = host.rth_kat_bvh_build(boxes.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(n), out_nodes.ctypes.data_as(C.POINTER(C.c_uint32)),
                                   C.byref(num_nodes), out_order.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert r == 0
        assert num_nodes.value == len(ref_nodes), n
        assert np.array_equal(out_order, ref_order), n
        assert np.array_equal(_mask_unused(out_nodes[:num_nodes.value]), _mask_unused(ref_nodes)), n


@pytest.fixture(scope="module")
def mesh_scene(built):
    return kat_io.mesh_fixture_scene()


def test_mesh_preprocessing_and_traversal_match_reference(mesh_scene):
    scene, mats, nt = mesh_scene
    d = scene.desc.contents
    raw = np.fromfile(os.path.join(kat_io.GOLDEN, "mesh_kat.bin"), dtype=np.uint32)
    num_nodes, num_tris = int(raw[0]), int(raw[1]); off = 2
    assert num_tris == nt == d.numTriangles
    ref_nodes = raw[off:off + 8 * num_nodes].reshape(num_nodes, 8); off += 8 * num_nodes
    ref_tris = raw[off:off + 13 * num_tris].reshape(num_tris, 13); off += 13 * num_tris
    # BVH nodes of the mesh
    assert d.numMeshNodes == num_nodes
    nodes = np.ctypeslib.as_array(C.cast(d.meshNodes, C.POINTER(C.c_uint32)), shape=(num_nodes, 8))
    assert np.array_equal(_mask_unused(nodes), _mask_unused(ref_nodes))
    # preprocessed triangles (v0, edge1, edge2) in leaf order + vertex indices; material ids are scene-global here
    tris = np.ctypeslib.as_array(C.cast(d.triangles, C.POINTER(C.c_uint32)), shape=(num_tris, 9))
    vidx = np.ctypeslib.as_array(C.cast(d.vertexIndices, C.POINTER(C.c_uint32)), shape=(num_tris, 4))
    assert np.array_equal(tris, ref_tris[:, :9])
    assert np.array_equal(vidx[:, :3], ref_tris[:, 9:12])
    # material ids: mesh-local in the reference, scene-global (interned) here -> must be a consistent injection
    pairs = set(zip(ref_tris[:, 12].tolist(), vidx[:, 3].tolist()))
    assert len(pairs) == len({a for a, _ in pairs}) == len({b for _, b in pairs})
    local_to_global = dict(pairs)
    # traversal + shading frames through the oracle
    num_rays = int(raw[off]); off += 1
    rec = raw[off:off + 26 * num_rays].reshape(num_rays, 26)
    rays = rec[:, :7].copy().view(np.float32)
    out = np.zeros((num_rays, 19), dtype=np.uint32)
    r = oracle_lib.lib().rto_kat_mesh(scene.desc, C.c_uint32(0), rays.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(num_rays),
                                      out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert r == 0
    exp = rec[:, 7:]
    assert (exp[:, 0] == 7).sum() > num_rays // 4          # the fixture exercises plenty of hits
    assert np.array_equal(out[:, :6], exp[:, :6])           # objectId, triangle, distance, u, v, any-hit: bit exact
    hit = exp[:, 0] == 7
    # frame: the tangent goes through FastNormalize3 (_mm_rsqrt_ps) in the reference -> tolerance; normal and uv exact
    assert np.array_equal(out[hit][:, 10:18], exp[hit][:, 10:18])
    t_ref, t_got = exp[hit][:, 6:9].copy().view(np.float32).astype(np.float64), out[hit][:, 6:9].copy().view(np.float32).astype(np.float64)
    assert np.all(np.abs(t_ref - t_got) <= 2.0 ** -11 * np.maximum(np.abs(t_ref), 1e-3))
    assert np.array_equal(out[hit][:, 18], np.array([local_to_global[int(m)] for m in exp[hit][:, 18]], dtype=np.uint32))
