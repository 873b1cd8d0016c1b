"""rtgpu_create_multi: ONE context over several devices (include/rtgpu.h) -- the frame's 64x64 tiles dealt to the devices, every call
fanned out, the read-back calls gathering the peers' tiles.  The bar is the 1-device context bit for bit: sums, secondary sums,
counters, post-processed front buffer, adaptive block errors.

The driver's GPU box has one device, so the shards here sit on the SAME device (a device index may repeat): everything but the xGMI
hop itself -- sharding, fan-out, per-device batching, the in-place gather kernel and the staged (hipMemcpyPeerAsync) gather -- runs
exactly as it would on several.  The last test uses two physical devices when the box has them."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VCM = 1     # RT_INTEGRATOR_VCM


class Shard(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("worldSize", C.c_uint32)]


def render(scene, camera, w, h, passes, devices=None, name="Path Tracer MIS", all_lights=False, depth=6, adaptive=False, counters=True):
    vp = ra.Viewport(w, h, seed=21, max_ray_depth=depth, light_sampling_all=all_lights)
    if adaptive:
        vp.set_adaptive(True, num_initial_passes=2, min_block_size=8, max_block_size=64, subdivision_treshold=0.05, convergence_treshold=0.01)
    vp.set_renderer(scene, name, devices=devices, intersection_counters=counters)
    vp.render(camera, passes)
    s, s2 = vp.sum_buffer(secondary=True)
    n = C.c_uint32(0)
    assert ra.rtgpu_lib().rtgpu_num_devices(vp.device_context(), C.byref(n)) == 0
    return dict(sum=s, secondary=s2, counters=vp.counters(), front=vp.front_buffer(), devices=int(n.value), progress=vp.progress() if adaptive else None, vp=vp)


def assert_same(a, b):
    assert np.array_equal(a["sum"].view(np.uint32), b["sum"].view(np.uint32))
    assert np.array_equal(a["secondary"].view(np.uint32), b["secondary"].view(np.uint32))
    assert np.array_equal(a["front"], b["front"])
    for k in ("numRays", "numPrimaryRays", "numShadowRays", "numShadowRaysHit"):
        assert a["counters"][k] == b["counters"][k], k


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_shards_of_one_context_reproduce_the_single_device_frame(built, shards):
    w, h = 328, 200      # 6 x 4 tiles, ragged right and bottom edges
    scene, camera = scenes.cornell_box(w / h)
    one = render(scene, camera, w, h, 11)
    many = render(scene, camera, w, h, 11, devices=[0] * shards)
    assert one["devices"] == 1 and many["devices"] == shards
    assert one["sum"].any()
    assert_same(one, many)
    # rtgpu_get_multi_info: which gather the context chose and why (shards of one device address each other: the in-place kernel), its timings
    info = ra.multi_info(many["vp"].device_context())
    assert info["numDevices"] == shards and info["devices"] == [0] * shards and all(info["peerAccess"])
    assert info["gatherMode"] == "peer-kernel" and info["gatherReason"] == "" and info["gathers"] >= 1 and info["lastGatherMs"] > 0.0
    single = ra.multi_info(one["vp"].device_context())
    assert single["numDevices"] == 1 and single["gatherMode"] == "none" and single["gathers"] == 0


def test_staged_gather_and_the_all_lights_strategy(built, monkeypatch):
    w, h = 256, 160
    scene, camera = scenes.cornell_box(w / h)
    one = render(scene, camera, w, h, 5, all_lights=True)
    monkeypatch.setenv("RTGPU_MULTI_STAGED", "1")     # hipMemcpyPeerAsync into staging buffers, then the same gather kernel
    many = render(scene, camera, w, h, 5, devices=[0, 0, 0], all_lights=True)
    assert_same(one, many)
    info = ra.multi_info(many["vp"].device_context())
    assert info["gatherMode"] == "staged-copy" and info["gatherReason"] == "RTGPU_MULTI_STAGED=1" and info["gathers"] >= 1


def test_mesh_scene_and_the_other_per_pixel_integrators(built):
    w, h = 320, 192
    scene, camera = scenes.sponza_class(w / h, target_triangles=20000)
    for name in ("Path Tracer MIS", "Path Tracer", "Debug"):
        one = render(scene, camera, w, h, 4, name=name)
        many = render(scene, camera, w, h, 4, devices=[0, 0], name=name)
        assert_same(one, many)
    # the library's default walk (intersection counters off: the 4-wide tree and its re-trace launches on every shard)
    one = render(scene, camera, w, h, 6, counters=False)
    many = render(scene, camera, w, h, 6, devices=[0, 0, 0], counters=False)
    assert_same(one, many)
    assert one["counters"]["numRetracedRays"] > 0 and many["counters"]["numRetracedRays"] == one["counters"]["numRetracedRays"]


def test_adaptive_rendering_over_shards(built):
    """Block errors are computed over the GATHERED frame and the active blocks are fanned out: the block list evolves identically."""
    w, h = 256, 256
    scene, camera = scenes.cornell_box(w / h)
    one = render(scene, camera, w, h, 9, adaptive=True)
    many = render(scene, camera, w, h, 9, devices=[0, 0, 0], adaptive=True)
    assert_same(one, many)
    assert one["progress"]["blocks"] == many["progress"]["blocks"]
    assert one["progress"]["averageError"] == many["progress"]["averageError"]
    assert len(one["progress"]["blocks"]) > 1


def test_whole_frame_integrators_stay_on_one_device_and_shards_are_refused(built):
    w, h = 128, 96
    scene, camera = scenes.cornell_box(w / h)
    vcm = render(scene, camera, w, h, 2, devices=[0, 0], name="VCM")     # the mirror gives VCM a single-device context
    assert vcm["devices"] == 1 and np.isfinite(vcm["sum"]).all() and vcm["sum"].any()
    lib = ra.rtgpu_lib()
    ctx = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.rtgpu_create_multi(devs, 2, C.byref(ctx)) == 0
    try:
        assert lib.rtgpu_set_shard(ctx, Shard(0, 2)) == -6                   # RTGPU_ERR_UNSUPPORTED
        assert lib.rtgpu_set_integrator(ctx, VCM, None) == -6
        assert b"single-device" in lib.rtgpu_last_error()
    finally:
        lib.rtgpu_destroy(ctx)
    bad = (C.c_int * 2)(0, 99)
    assert lib.rtgpu_create_multi(bad, 2, C.byref(ctx)) == -1
    assert lib.rtgpu_create_multi(devs, 17, C.byref(ctx)) == -1


def test_rt_demo_scales_through_the_environment(built, tmp_path):
    """The headless Demo over `RTGPU_DEVICES`: same BMP as on one device."""
    exe = os.path.join(ROOT, "raytracer_amd", "lib", "rt_demo")
    scene = os.path.join(ROOT, "tests", "golden", "obj", "scene.json")
    outs = []
    for devices in (None, "0,0,0,0"):
        out = str(tmp_path / ("multi.bmp" if devices else "one.bmp"))
        env = dict(os.environ)
        env.pop("RTGPU_DEVICES", None)
        if devices:
            env["RTGPU_DEVICES"] = devices
        r = subprocess.run([exe, "-s", scene, "--data", os.path.dirname(scene) + "/", "--width", "200", "--height", "136", "--passes", "6", "--depth", "5", "--seed", "11", "--output", out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 200 * 136 * 3


def test_sharded_viewports_reproduce_the_unsharded_frame(built):
    """The one-process-per-GPU front end (bench.py --gpus N): separate viewports with the same seed, each owning a shard.  Their frames
    add up to the unsharded viewport's frame bit for bit -- set_shard must not disturb the sample sequence -- with the library's default
    walk (intersection counters off: the 4-wide tree) and with the binary one."""
    w, h, passes = 320, 200, 5
    scene, camera = scenes.sponza_class(w / h, 20000)

    def run(shard=None, counters=False):
        vp = ra.Viewport(w, h, seed=77, max_ray_depth=6)
        vp.set_renderer(scene, intersection_counters=counters)
        if shard:
            vp.set_shard(*shard)
        vp.render(camera, passes)
        return vp.sum_buffer(), vp.counters()

    for counters in (False, True):
        whole, cw = run(counters=counters)
        parts = [run((r, 3), counters) for r in range(3)]
        total = parts[0][0] + parts[1][0] + parts[2][0]        # disjoint support: adding zeros is exact
        assert np.array_equal(total.view(np.uint32), whole.view(np.uint32))
        for k in ("numRays", "numShadowRays", "numShadowRaysHit", "numPrimaryRays"):
            assert sum(p[1][k] for p in parts) == cw[k], k
    assert cw["numRayBoxTests"] > 0


def test_two_physical_devices(built):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one device on this box: the xGMI hop of the gather is covered by the driver's multi-GPU run")
    w, h = 328, 200
    scene, camera = scenes.cornell_box(w / h)
    one = render(scene, camera, w, h, 11)
    two = render(scene, camera, w, h, 11, devices=[0, 1])
    assert_same(one, two)
