"""Diagnostic: first pass / pixels at which the VCM sum buffer of the caustics scene stops being finite."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import raytracer_amd as ra
from bench_vcm import caustics_scene

w, h = 1920, 1080
scene, camera = caustics_scene(ra, w / h)
vp = ra.Viewport(w, h, seed=20260928)
vp.set_renderer(scene, name="VCM")
kw = {}
if len(sys.argv) > 2:
    kw = json.loads(sys.argv[2])
    vp.set_vcm(**kw)
out = {"found": False, "settings": kw}
for i in range(int(sys.argv[1])):
    vp.render_pass_with(vp.next_pass_params(camera))
    img = vp.sum_buffer()
    if not np.isfinite(img).all():
        ys, xs = np.nonzero(~np.isfinite(img).all(axis=2))
        out = {"found": True, "pass": i, "count": int(len(xs)), "pixels": [[int(x), int(y)] for x, y in zip(xs[:16], ys[:16])],
               "values": [[float(v) for v in img[y, x]] for x, y in zip(xs[:4], ys[:4])], "settings": kw}
        break
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/vcm_nan_probe.json", "w").write(json.dumps(out))
