"""Diagnostic: oracle-only replay of one pass of the caustics scene + hashes of every input, to compare two machines."""
import ctypes as C
import hashlib
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
target, px, py = 56, 700, 560
scene, camera = caustics_scene(ra, w / h)
desc = scene.desc
bn = ra.load_blue_noise(); desc.contents.blueNoise = bn.ctypes.data
d = desc.contents
def hbytes(ptr, n):
    return hashlib.md5(C.string_at(ptr, n)).hexdigest()[:12] if ptr and n else "-"
print("objects", hbytes(d.objects, d.numObjects * C.sizeof(ra.RtObject)), "lights", hbytes(d.lights, d.numLights * C.sizeof(ra.RtLight)),
      "materials", hbytes(d.materials, d.numMaterials * C.sizeof(ra.RtMaterial)), "topNodes", hbytes(d.topNodes, d.numTopNodes * 32), "bn", hashlib.md5(bn.tobytes()).hexdigest()[:12])
vp = ra.Viewport(w, h, seed=20260928)
params = [vp.next_pass_params(camera) for _ in range(target + 1)]
p = params[target]
print("camera", hashlib.md5(bytes(p.camera)).hexdigest()[:12], "seed", hbytes(p.seed, 4 * p.numDimensions), "dims", p.numDimensions, "key", p.rngKey[0], p.rngKey[1], "offset", p.sampleOffset[0], p.sampleOffset[1])
tile = (py // 64) * ((w + 63) // 64) + px // 64
vcm = oracle_lib.Vcm(use_vertex_merging=False)
vcm.passes = target
cam = np.zeros((h, w, 3), np.float32); light = np.zeros((h, w, 3), np.float32)
vcm.render_pass(desc, p, w, h, cam, None, light, shard=(tile % 600, 600))
print("oracle cam pixel", cam[py, px], "tile hash", hashlib.md5(cam[py // 64 * 64:py // 64 * 64 + 64, px // 64 * 64:px // 64 * 64 + 64].tobytes()).hexdigest()[:12])
