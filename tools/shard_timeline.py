"""Where the time of a short run on a small shard goes: host queueing, device work, read-back (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import raytracer_amd as ra
from raytracer_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w, h = 1920, 1080
scene, camera = scenes.sponza_class(w / h)
vp = ra.Viewport(w, h, seed=77, max_ray_depth=8)
vp.set_renderer(scene, intersection_counters=False)
if n > 1:
    vp.set_shard(0, n)
lib, host = ra.rtgpu_lib(), ra.host_lib()
ctx = vp.device_context()
vp.render(camera, 5); lib.rtgpu_synchronize(ctx); host.rth_viewport_fetch_sum(vp._h)
for rep in range(3):
    t0 = time.perf_counter()
    vp.render(camera, steps)
    t1 = time.perf_counter()
    lib.rtgpu_synchronize(ctx)
    t2 = time.perf_counter()
    host.rth_viewport_fetch_sum(vp._h)
    t3 = time.perf_counter()
    print("shard 1/%d, %d passes: queueing %.2f ms, wait for the device %.2f ms, read-back %.2f ms, total %.2f ms = %.3f ms per pass" % (
        n, steps, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), 1e3 * (t3 - t0) / steps))
