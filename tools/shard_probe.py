"""Sharded vs unsharded frame with the library's default walk (intersection counters off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import scenes
w, h, passes = 640, 360, 8
scene, camera = scenes.sponza_class(w / h, 20000)
def run(shard=None, counters=False):
    vp = ra.Viewport(w, h, seed=77, max_ray_depth=8)
    vp.set_renderer(scene, intersection_counters=counters)
    if shard: vp.set_shard(*shard)
    vp.render(camera, passes)
    return vp.sum_buffer(), vp.counters()
a, ca = run()
b, cb = run()
print("unsharded twice identical:", np.array_equal(a.view(np.uint32), b.view(np.uint32)), ca["numRays"], cb["numRays"], ca["numRetracedRays"])
s0, c0 = run((0, 2)); s1, c1 = run((1, 2))
s = s0 + s1
print("2 shards vs whole:", np.array_equal(s.view(np.uint32), a.view(np.uint32)), c0["numRays"] + c1["numRays"], ca["numRays"], "differing pixels", int(np.count_nonzero((s != a).any(axis=2))))
e, ce = run(counters=True)
print("counters on (binary walk) vs default:", np.array_equal(e.view(np.uint32), a.view(np.uint32)), ce["numRays"], ca["numRays"], "differing pixels", int(np.count_nonzero((e != a).any(axis=2))))
os.environ["RTGPU_WIDE"] = "0"
f, cf = run()
print("RTGPU_WIDE=0 vs default:", np.array_equal(f.view(np.uint32), a.view(np.uint32)), cf["numRays"])
