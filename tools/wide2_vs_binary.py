"""Exactness check of the two-level 4-wide walk (k_trace_wide2) ON the device: the same passes with RTGPU_WIDE2=1 and =0 (the binary walk),
640x480 x 16 passes per scene and depth; prints the number of differing pixels and the ray counters of both.  usage: python tools/wide2_vs_binary.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import scenes
import scene_zoo
w, h = 640, 480
for name, make in (("cornell", scenes.cornell_box), ("mesh_scene", lambda a: scene_zoo.mesh_scene(a, triangles=20000)), ("zoo", scene_zoo.all_lights_scene)):
    scene, camera = make(w / h)
    for depth in (1, 6):
        imgs = {}
        params = None
        for wide2 in ("1", "0"):
            os.environ["RTGPU_WIDE2"] = wide2
            vp = ra.Viewport(w, h, seed=7, max_ray_depth=depth)
            vp.set_renderer(scene)
            if params is None:
                params = [vp.next_pass_params(camera) for _ in range(16)]
            for p in params:
                vp.render_pass_with(p)
            imgs[wide2] = (vp.sum_buffer().copy(), vp.counters())
        a, b = imgs["1"][0], imgs["0"][0]
        bad = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())
        ca, cb = imgs["1"][1], imgs["0"][1]
        print(name, "depth", depth, "differing pixels", bad, {k: (ca[k], cb[k]) for k in ("numRays", "numShadowRays", "numShadowRaysHit", "numRetracedRays")}, flush=True)
