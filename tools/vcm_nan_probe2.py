"""Diagnostic: render ONE given pass of the caustics scene (merging off: no cross-pass state) and compare a tile subset with the oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib
import raytracer_amd as ra
from bench_vcm import caustics_scene

w, h = 1920, 1080
target, px, py = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene, camera = caustics_scene(ra, w / h)
desc = scene.desc
bn = ra.load_blue_noise(); desc.contents.blueNoise = bn.ctypes.data
vp = ra.Viewport(w, h, seed=20260928)
vp.set_renderer(scene, name="VCM")
vp.set_vcm(use_vertex_merging=False)
params = [vp.next_pass_params(camera) for _ in range(target + 1)]
p = params[target]
import ctypes as C, hashlib
def hbytes(ptr, n):
    return hashlib.md5(C.string_at(ptr, n)).hexdigest()[:12] if ptr and n else "-"
print("before: camera", hashlib.md5(bytes(p.camera)).hexdigest()[:12], "seed", hbytes(p.seed, 4 * p.numDimensions), "dims", p.numDimensions, "key", p.rngKey[0], p.rngKey[1], "offset", p.sampleOffset[0], p.sampleOffset[1], "pass", p.passIndex)
import json
json.dump({"camera": list(bytes(p.camera)), "seed": [int(p.seed[i]) for i in range(p.numDimensions)], "key": [int(p.rngKey[0]), int(p.rngKey[1])],
           "offset": [float(p.sampleOffset[0]), float(p.sampleOffset[1])], "passIndex": int(p.passIndex), "useBlueNoise": int(p.useBlueNoise),
           "numDimensions": int(p.numDimensions)}, open("gpurun_out/vcm_nan_params.json", "w"))
vp.render_pass_with(p)
print("after:  camera", hashlib.md5(bytes(p.camera)).hexdigest()[:12], "seed", hbytes(p.seed, 4 * p.numDimensions), "dims", p.numDimensions, "key", p.rngKey[0], p.rngKey[1], "offset", p.sampleOffset[0], p.sampleOffset[1], "pass", p.passIndex)
img = vp.sum_buffer()
print("gpu pixel", img[py, px], "non-finite pixels", int((~np.isfinite(img).all(axis=2)).sum()))
tile = (py // 64) * ((w + 63) // 64) + px // 64
world = 600
vcm = oracle_lib.Vcm(use_vertex_merging=False)
vcm.passes = target
cam = np.zeros((h, w, 3), np.float32); light = np.zeros((h, w, 3), np.float32)
vcm.render_pass(desc, p, w, h, cam, None, light, shard=(tile % world, world))
print("oracle cam pixel", cam[py, px], "oracle light pixel (own tile's light paths only)", light[py, px])
ys, xs = np.nonzero(~np.isfinite(light).all(axis=2))
print("oracle non-finite light pixels from this tile's paths:", list(zip(xs[:5], ys[:5])))
