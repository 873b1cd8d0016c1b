"""SURVEY 8(f) row 3 -- scene ingestion in the C++ host mirror (raytracer_amd/host/Demo/): helpers::LoadMesh (OBJ / MTL /
BMP, vertex de-duplication, Lengyel tangents) against the REFERENCE's own Demo/MeshLoader.cpp + vendored tinyobjloader
(tests/golden/obj_mesh_kat.bin, produced by oracle/ref_harness from tests/golden/obj/fixture.obj), and
helpers::LoadScene (JSON) against the Python scene builder on the Cornell box."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import kat_io
import raytracer_amd as ra
from raytracer_amd import scenes

OBJ_DIR = os.path.join(kat_io.GOLDEN, "obj")


def _mask_unused(nodes):
    """Node 1 is never written (root at 0, child pairs from 2) and a leaf's splitAxis is uninitialised in the reference."""
    nodes = nodes.copy()
    if len(nodes) > 1:
        nodes[1] = 0
    leaf = (nodes[:, 7] & 0x3FFFFFFF) != 0
    nodes[leaf, 7] &= 0x3FFFFFFF
    return nodes


def _desc_arrays(d):
    return dict(
        nodes=np.ctypeslib.as_array(C.cast(d.meshNodes, C.POINTER(C.c_uint32)), shape=(d.numMeshNodes, 8)).copy(),
        tris=np.ctypeslib.as_array(C.cast(d.triangles, C.POINTER(C.c_uint32)), shape=(d.numTriangles, 9)).copy(),
        vidx=np.ctypeslib.as_array(C.cast(d.vertexIndices, C.POINTER(C.c_uint32)), shape=(d.numTriangles, 4)).copy(),
        shading=np.ctypeslib.as_array(C.cast(d.vertexShading, C.POINTER(C.c_uint32)), shape=(d.numVertices, 8)).copy())


def test_obj_mesh_ingestion_matches_the_reference_loader(built, tmp_path):
    scene_file = tmp_path / "mesh.json"
    scene_file.write_text(json.dumps({"objects": [{"type": "mesh", "path": "fixture.obj", "scale": 1.25}]}))
    scene = ra.Scene().load_json(scene_file, data_path=OBJ_DIR + "/")
    scene.build()
    d = scene.desc.contents

    raw = np.fromfile(os.path.join(kat_io.GOLDEN, "obj_mesh_kat.bin"), dtype=np.uint32)
    num_nodes, num_tris, num_mats = (int(v) for v in raw[:3]); off = 3
    ref_nodes = raw[off:off + 8 * num_nodes].reshape(num_nodes, 8); off += 8 * num_nodes
    ref_tris = raw[off:off + 37 * num_tris].reshape(num_tris, 37); off += 37 * num_tris
    ref_mats = raw[off:off + 10 * num_mats].reshape(num_mats, 10); off += 10 * num_mats
    assert off == raw.size and num_tris > 200 and num_mats == 3

    a = _desc_arrays(d)
    assert d.numMeshNodes == num_nodes and d.numTriangles == num_tris
    # BVH nodes (built over the loader's triangles), preprocessed triangles in leaf order, vertex indices: bit identical
    assert np.array_equal(_mask_unused(a["nodes"]), _mask_unused(ref_nodes))
    assert np.array_equal(a["tris"], ref_tris[:, :9])
    assert np.array_equal(a["vidx"][:, :3], ref_tris[:, 9:12])
    # normals, Lengyel tangents and texture coordinates of the three vertices of every triangle: bit identical
    for k in range(3):
        got = a["shading"][a["vidx"][:, k]]
        assert np.array_equal(got, ref_tris[:, 13 + 8 * k:21 + 8 * k]), "vertex %d of some triangle differs" % k
    # materials: MTL colours, LoadMaterial's fixed roughness, the diffuse BMP texture; -1 (unknown usemtl) -> default material
    mats = np.ctypeslib.as_array(C.cast(d.materials, C.POINTER(C.c_uint32)), shape=(d.numMaterials, C.sizeof(ra.RtMaterial) // 4))
    local_to_global = {}
    for local, glob in set(zip(ref_tris[:, 12].tolist(), a["vidx"][:, 3].tolist())):
        assert local_to_global.setdefault(local, glob) == glob
    assert 0xFFFFFFFF in local_to_global and local_to_global[0xFFFFFFFF] == 0xFFFFFFFF      # falls back to the object's default material
    for local in range(num_mats):
        if local not in local_to_global:
            continue
        m = ra.RtMaterial.from_buffer_copy(mats[local_to_global[local]].tobytes())
        ref = ref_mats[local]
        assert np.array_equal(np.array(m.baseColor[:3], dtype=np.float32).view(np.uint32), ref[0:3])
        assert np.array_equal(np.array(m.emission[:3], dtype=np.float32).view(np.uint32), ref[3:6])
        assert np.float32(m.roughness).view(np.uint32) == ref[6] and m.bsdf == 1
        assert (m.baseColorTexture != ra.RT_NO_TEXTURE) == bool(ref[7])
        if ref[7]:
            t = d.textures[m.baseColorTexture]
            assert (t.width, t.height, t.format, t.linearSpace) == (int(ref[8]), int(ref[9]), 3, 0)   # 24-bit BMP: B8G8R8, sRGB


def test_tinyobj_number_syntax(built):
    """tryParseDouble of tinyobjloader 1.4.0 (restated in host/Demo/ObjReader.cpp): accepted and rejected forms."""
    h = ra.host_lib()
    out = C.c_double()
    for text, value in (("1", 1.0), ("-2.5", -2.5), ("+.5", None), ("3.", 3.0), ("1e3", 1000.0), ("1.5E-2", 0.015), ("-0.000001", -1e-6), ("12.125e+1", 121.25)):
        r = h.rth_kat_parse_double(text.encode(), C.byref(out))
        if value is None:
            assert r != 0, text
        else:
            assert r == 0 and abs(out.value - value) <= 1e-15 * max(1.0, abs(value)), (text, out.value)
    assert h.rth_kat_parse_double(b"abc", C.byref(out)) != 0 and h.rth_kat_parse_double(b"1e", C.byref(out)) != 0


def _as_json(value):
    """json text in which every float keeps a fractional part (rapidjson's IsFloat() rejects integer tokens)."""
    return json.dumps(value)


def test_json_scene_loader_builds_the_same_scene_as_the_python_builder(built, tmp_path):
    path = tmp_path / "cornell_box.json"
    path.write_text(_as_json(scenes.CORNELL_BOX))
    cam = ra.Camera()
    loaded = ra.Scene().load_json(path, camera=cam)
    loaded.build()
    expected, expected_cam = scenes.load_json_scene(scenes.CORNELL_BOX, aspect=1.0)
    a, b = loaded.desc.contents, expected.desc.contents
    for field in ("numObjects", "numTopNodes", "numLights", "numGlobalLights", "numMaterials", "numMeshes", "numTextures"):
        assert getattr(a, field) == getattr(b, field), field
    for ptr, count, ty in (("objects", a.numObjects, ra.RtObject), ("lights", a.numLights, ra.RtLight), ("materials", a.numMaterials, ra.RtMaterial),
                           ("topNodes", a.numTopNodes, ra.RtNode)):
        size = C.sizeof(ty) * count
        assert C.string_at(C.cast(getattr(a, ptr), C.c_void_p), size) == C.string_at(C.cast(getattr(b, ptr), C.c_void_p), size), ptr
    ca, cb = ra.RtCamera(), ra.RtCamera()
    assert ra.host_lib().rth_camera_desc(cam._h, C.byref(ca)) == 0 and ra.host_lib().rth_camera_desc(expected_cam._h, C.byref(cb)) == 0
    assert bytes(ca) == bytes(cb)


def test_json_loader_rejects_what_the_reference_rejects(built, tmp_path):
    bad = tmp_path / "bad.json"
    # an integer token where TryParseFloat demands a real (rapidjson IsFloat), SceneLoader.cpp:124-145
    bad.write_text('{"objects": [{"type": "sphere", "radius": 1}]}')
    with pytest.raises(ValueError):
        ra.Scene().load_json(bad)
    bad.write_text('{"objects": [{"type": "sphere", "radius": 1.0, "material": "nope"}]}')
    with pytest.raises(ValueError):
        ra.Scene().load_json(bad)
    bad.write_text('{"objects": [{"type": "sphere", "radius": 1.0}], "lights": [{"type": "area", "color": [1.0, 1.0, 1.0]}]}')
    with pytest.raises(ValueError):
        ra.Scene().load_json(bad)
    bad.write_text('{"objects": [')
    with pytest.raises(ValueError):
        ra.Scene().load_json(bad)


def test_malformed_inputs_fail_to_load_instead_of_reading_out_of_bounds(built, tmp_path):
    scene_file = tmp_path / "mesh.json"
    scene_file.write_text(json.dumps({"objects": [{"type": "mesh", "path": "broken.obj", "scale": 1.0}]}))
    # a face that references vertex 9 of a file that defines 3: a load error
    (tmp_path / "broken.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(ValueError):
        ra.Scene().load_json(scene_file, data_path=str(tmp_path) + "/")
    # normal / uv indices beyond what the file defines count as missing: the face normal and zero uv are used
    (tmp_path / "broken.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0.5 0.5\nf 1/7/5 2/1/1 3/1/1\n")
    scene = ra.Scene().load_json(scene_file, data_path=str(tmp_path) + "/")
    scene.build()
    d = scene.desc.contents
    assert d.numTriangles == 1
    shading = np.ctypeslib.as_array(C.cast(d.vertexShading, C.POINTER(C.c_float)), shape=(d.numVertices, 8))
    assert np.isfinite(shading).all()
    # 100 000 nested arrays: an error, not a stack overflow
    deep = tmp_path / "deep.json"
    deep.write_text('{"objects": ' + "[" * 100000 + "]" * 100000 + "}")
    with pytest.raises(ValueError):
        ra.Scene().load_json(deep)


def test_camera_world_to_screen_matches_reference(built):
    """Camera::SetPerspective's mWorldToScreen (FastInverseNoScale x MakePerspective, with the reference's operation
    order and its stray w lanes) recomputed by the host mirror from the camera_film.kat inputs: bit-exact."""
    import ctypes as C
    import kat_io
    _, inputs, _ = kat_io.load_kat("camera_film.kat")
    cw = C.sizeof(ra.RtCamera) // 4
    h = ra.host_lib()
    h.rth_kat_world_to_screen.restype = None
    for row in inputs[:256]:
        cam = ra.RtCamera.from_buffer_copy(row[:cw].tobytes())
        out = (C.c_float * 16)()
        h.rth_kat_world_to_screen(cam.localToWorld, C.c_float(cam.aspectRatio), C.c_float(cam.tanHalfFoV), out)
        assert np.array_equal(np.frombuffer(out, np.uint32), np.frombuffer(cam.worldToScreen, np.uint32))


def test_dds_loader_matches_the_reference(built):
    """Bitmap::LoadDDS: 45 header variants (tests/golden/dds/, own data) against what the reference's Bitmap::Load made of the same
    files (dds_kat.bin): texel format, colour space, size, payload -- and the same three files refused (DXT5, BC7, truncated)."""
    import ctypes as C
    import glob
    import kat_io
    raw = np.fromfile(os.path.join(kat_io.GOLDEN, "dds_kat.bin"), dtype=np.uint32)
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(kat_io.GOLDEN, "dds", "*.dds")))
    assert raw[0] == len(names) == 45
    h = ra.host_lib()
    refused = 0
    for k, name in enumerate(names):
        rec = raw[1 + 8 * k:1 + 8 * (k + 1)]
        fnv = 2166136261
        for ch in name.encode():
            fnv = ((fnv ^ ch) * 16777619) & 0xFFFFFFFF
        assert rec[0] == fnv, name
        out = (C.c_uint32 * 6)()
        r = h.rth_kat_load_bitmap(os.path.join(kat_io.GOLDEN, "dds", name).encode(), out)
        assert (r == 0) == bool(rec[1]), name
        if r == 0:
            assert list(out) == [int(v) for v in rec[2:8]], (name, list(out), rec[2:8])
        else:
            refused += 1
    assert refused == 3


def test_bmp_loader_matches_the_reference(built):
    """Bitmap::LoadBMP: 24-bit, 8-bit grey, 8-bit with a colour table of 2 / 16 / 256 entries (B8G8R8A8_UNorm_Palette: palette and
    index bytes), padded rows -- against what the reference's Bitmap::Load made of the same files (tests/golden/bmp/, own data;
    bmp_kat.bin) -- and the same four files refused (4-bit, RLE, truncated, two planes)."""
    import ctypes as C
    import glob
    import kat_io
    raw = np.fromfile(os.path.join(kat_io.GOLDEN, "bmp_kat.bin"), dtype=np.uint32)
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(kat_io.GOLDEN, "bmp", "*.bmp")))
    assert raw[0] == len(names) == 11
    h = ra.host_lib()
    refused, palettized = 0, 0
    for k, name in enumerate(names):
        rec = raw[1 + 10 * k:1 + 10 * (k + 1)]
        fnv = 2166136261
        for ch in name.encode():
            fnv = ((fnv ^ ch) * 16777619) & 0xFFFFFFFF
        assert rec[0] == fnv, name
        out = (C.c_uint32 * 6)(); pal = (C.c_uint32 * 2)()
        path = os.path.join(kat_io.GOLDEN, "bmp", name).encode()
        r = h.rth_kat_load_bitmap(path, out)
        assert (r == 0) == bool(rec[1]), name
        if r == 0:
            assert list(out) == [int(v) for v in rec[2:8]], (name, list(out), rec[2:8])
            assert h.rth_kat_load_bitmap_palette(path, pal) == 0 and list(pal) == [int(v) for v in rec[8:10]], (name, list(pal), rec[8:10])
            palettized += int(rec[8]) > 0
        else:
            refused += 1
    assert refused == 4 and palettized == 3
