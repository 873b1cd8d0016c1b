"""Tuning probe: throughput of the PathTracerMIS pipeline when ALL of the chip works on one eighth of the frame (a horizontal
band) -- an upper estimate of what an XCD-partitioned work distribution (each XCD's L2 serving one image region) could reach."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
from raytracer_amd import scenes

w, h = 1920, 1080
scene, camera = scenes.sponza_class(w / h)
lib = ra.rtgpu_lib()


def run(blocks, passes):
    vp = ra.Viewport(w, h, seed=20260928, max_ray_depth=8)
    vp.set_renderer(scene)
    ctx = vp.device_context()
    lib.rtgpu_set_intersection_counters(ctx, 0)
    if blocks:
        arr = (ra.RtBlock * len(blocks))(*[ra.RtBlock(*b) for b in blocks])
        assert lib.rtgpu_set_active_blocks(ctx, len(blocks), arr) == 0, lib.rtgpu_last_error()
    vp.render(camera, 16)
    lib.rtgpu_synchronize(ctx)
    c0 = vp.counters()
    t0 = time.perf_counter()
    vp.render(camera, passes)
    lib.rtgpu_synchronize(ctx)
    dt = time.perf_counter() - t0
    c1 = vp.counters()
    return (c1["numRays"] - c0["numRays"]) / dt / 1e6


print("full frame          %.0f Msamples/s" % run(None, 32))
tot = 0.0
for k in range(8):
    v = run([(0, w, 135 * k, 135 * (k + 1))], 128)
    tot += 1.0 / v
    print("band %d (rows %4d-%4d) %.0f Msamples/s" % (k, 135 * k, 135 * (k + 1), v))
print("harmonic mean of the bands %.0f Msamples/s" % (8.0 / tot))
for k in range(2):
    v = run([(240 * k * 4, 240 * (k * 4 + 1), 0, h)], 128)
    print("column band %d %.0f Msamples/s" % (k, v))
