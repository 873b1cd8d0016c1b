"""Path-level parity with the reference's own integrator: tests/golden/ref_paths/*.bin hold every vertex of every path of one pass as
PathTracerMIS::RenderPixel's own debugging hook recorded them (PathDebugData; tests/golden/make_ref_paths_fixtures.py).  The oracle's
renderPixel records the same fields at the same two places (oracle/rto_core.h: pathDumpRecord) for every pixel of the same pass.

What must hold: every path is found by its primary ray (bit-exact direction), has the same number of vertices, and agrees bit for bit on
every meaningful field -- ray, hit ids, distance, (u, v) of triangle hits, shading frame, texture coordinates, throughput, sampled BSDF
event -- up to the first vertex that depends on one of the two approximate instructions the reference bakes in (_mm_rsqrt_ps in
FastNormalize3: sphere frames, mesh tangents; _mm_rcp_ss in FastDivide: MIS weights), which are exact operations in the oracle and on
the device.  A mesh of flat-shaded boxes under an area light has no such site on the recorded fields: all of its paths are identical.
(The device is bit-identical to the oracle: tests/test_gpu_parity.py.)"""
import os
import struct

import numpy as np
import pytest

import oracle_lib
import ref_scenes
import raytracer_amd as ra

HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = ["ox", "oy", "oz", "dx", "dy", "dz", "object", "subObject", "distance", "u", "v", "px", "py", "pz", "nx", "ny", "nz", "tx", "ty", "tz",
          "uvx", "uvy", "tpx", "tpy", "tpz", "tpw", "bsdfEvent", "pad"]


def load(name):
    raw = open(os.path.join(HERE, "golden", "ref_paths", name + ".bin"), "rb").read()
    magic, w, h, depth, sampling_all, dims, n, _ = struct.unpack("<8I", raw[:32])
    assert magic == 0x31565052
    v = np.frombuffer(raw, dtype=np.float32, offset=32).reshape(n, 28)
    camera_position = v[0, :3].view(np.uint32)
    starts = np.nonzero((v[:, :3].view(np.uint32) == camera_position).all(axis=1))[0]     # a path starts where the ray leaves the camera
    paths = {}
    for i, s in enumerate(starts):
        paths[v[s, 3:6].tobytes()] = v[s:(starts[i + 1] if i + 1 < len(starts) else n)]
    assert len(paths) == w * h
    return dict(w=w, h=h, depth=depth, sampling_all=bool(sampling_all), dims=dims, paths=paths)


def compare(name):
    fx = load(name)
    w, h = fx["w"], fx["h"]
    scene, camera = ref_scenes.FIXTURES[name][0](w / h)
    desc = scene.desc.contents
    bn = ra.load_blue_noise()
    desc.blueNoise = bn.ctypes.data
    vp = ra.Viewport(w, h, seed=ref_scenes.SEED, max_ray_depth=fx["depth"], dimensions=fx["dims"], light_sampling_all=fx["sampling_all"])
    vp.reset()        # SetRenderer + Reset of the reference's callers
    p = vp.next_pass_params(camera)
    stats = dict(paths=0, same_length=0, identical=0, close=0, vertices=0, identical_vertices=0, first_difference={})
    for y in range(h):
        for x in range(w):
            mine = oracle_lib.render_pixel_paths(scene.desc, p, w, h, x, y)
            ref = fx["paths"].get(mine[0, 3:6].tobytes())
            assert ref is not None, "no reference path starts with the primary ray of pixel (%d, %d)" % (x, y)
            stats["paths"] += 1
            if len(ref) != len(mine):
                continue
            stats["same_length"] += 1
            d = mine.view(np.uint32) != ref.view(np.uint32)
            d[:, 27] = False
            d[-1, 26] = False                                   # the reference leaves bsdfEvent of a path's last record unset
            miss = ref[:, 6].view(np.uint32) == 0xFFFFFFFF
            d[miss, 7] = False; d[miss, 9:22] = False           # nothing was hit: sub-object, (u, v) and the shading data are whatever they were
            mesh = np.array([(not m) and desc.objects[int(o)].objectKind == 0 and desc.objects[int(o)].shapeKind == 3
                             for m, o in zip(miss, ref[:, 6].view(np.uint32))])
            d[~mesh, 9:11] = False                              # (u, v) are written by triangle hits only
            stats["vertices"] += len(ref)
            bad = np.nonzero(d.any(axis=1))[0]
            stats["identical_vertices"] += len(ref) if len(bad) == 0 else int(bad[0])
            if len(bad) == 0:
                stats["identical"] += 1
            else:
                f = FIELDS[int(np.nonzero(d[bad[0]])[0][0])]
                stats["first_difference"][f] = stats["first_difference"].get(f, 0) + 1
            # "close": same hit ids and sampled events everywhere, every compared float within 1e-3 (relative + absolute)
            floats = d & np.isfinite(mine) & np.isfinite(ref) & ~np.isin(np.arange(28), (6, 7, 26))[None, :]
            rel = np.abs(mine[floats] - ref[floats]) / (1e-3 + np.abs(ref[floats]))
            if not d[:, (6, 7, 26)].any() and not (d & ~floats).any() and (rel.size == 0 or rel.max() < 1e-3):
                stats["close"] += 1
    return stats


def test_flat_shaded_mesh_paths_are_identical_to_the_reference(built):
    s = compare("box_mesh")
    assert s["same_length"] == s["paths"] == s["identical"], s


@pytest.mark.parametrize("name,first_sites,identical,close,vertices", [("cornell", {"nx", "ny", "nz"}, 0.75, 0.85, 0.75),
                                                                       ("mesh_2k_all", {"tx", "ty", "tz"}, 0.5, 0.99, 0.55),
                                                                       ("mesh_single", {"tx", "ty", "tz"}, 0.48, 0.99, 0.55)])
def test_paths_agree_up_to_the_first_approximate_instruction(built, name, first_sites, identical, close, vertices):
    """Spheres (Cornell box) and interpolated mesh tangents are where _mm_rsqrt_ps enters: the first differing field of a path that
    differs is a normal / tangent component there (or a throughput that a _mm_rcp_ss MIS weight went into), never a hit id, a distance,
    a position or a sampled event; before that vertex everything is bit-identical."""
    s = compare(name)
    assert s["same_length"] >= 0.99 * s["paths"], s
    assert s["identical"] >= identical * s["paths"], s          # (a tangent's last bits differ at the first vertex of half the mesh paths)
    assert s["close"] >= close * s["paths"], s                  # same ids and events everywhere, floats within 1e-3
    assert s["identical_vertices"] >= vertices * s["vertices"], s
    assert set(s["first_difference"]) <= first_sites | {"tpx", "tpy", "tpz", "tpw"}, s
    print(name, s)


@pytest.mark.parametrize("name", ["box_mesh", "cornell", "mesh_2k_all", "mesh_single"])
def test_with_the_two_approximate_instructions_every_recorded_vertex_is_the_reference_s(built, name):
    """The same comparison with the oracle in x86 approximation mode (FastDivide through _mm_rcp_ss, FastNormalize3 through _mm_rsqrt_ps, like the
    reference; oracle/rto_math.h): no "up to the first approximate instruction" any more -- EVERY path has the reference's vertex count and EVERY
    recorded field of EVERY vertex its bits (sphere normals and mesh tangents included).  Skipped on a CPU with other approximation tables."""
    from test_reference_images import FIXTURE_CPU_SIGNATURE
    ok, signature = oracle_lib.set_x86_approximations(True)
    try:
        if not ok or signature != FIXTURE_CPU_SIGNATURE:
            pytest.skip("this CPU's _mm_rcp_ss / _mm_rsqrt_ps tables are not those of the CPU the fixtures were rendered on")
        s = compare(name)
    finally:
        oracle_lib.set_x86_approximations(False)
    assert s["same_length"] == s["paths"] == s["identical"] and s["identical_vertices"] == s["vertices"], s
