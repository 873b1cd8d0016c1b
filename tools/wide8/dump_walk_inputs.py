"""Step-model inputs for the 8-wide walk study (round 6, review item 1): the benchmark mesh's BINARY tree (the reference's BVH::Node array as
uploaded), its triangles in leaf order, and a sample of the rays a depth-8 PathTracerMIS pass actually traces -- the closest-hit rays of every
path of a stratified pixel sample, as the CPU oracle's path recording gives them (origin, direction, hit distance), plus the hit points'
frames from which tools/wide8/walk_model.cpp synthesises next-event rays (towards the sun / over the hemisphere).
  python tools/wide8/dump_walk_inputs.py [out_dir=/tmp/walk_model] [pixel stride=9] [triangles=262144]
Test infrastructure (uses the oracle); nothing here is on the product path."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as ra          # noqa: E402
from raytracer_amd import scenes    # noqa: E402
import oracle_lib                   # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/walk_model"
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    tris = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
    os.makedirs(out, exist_ok=True)
    w, h, depth = 1920, 1080, 8
    scene, camera = scenes.sponza_class(w / h, tris)
    d = scene.desc.contents
    bn = ra.load_blue_noise()
    d.blueNoise = bn.ctypes.data
    mesh = d.meshes[0]
    nodes = np.ctypeslib.as_array(C.cast(d.meshNodes, C.POINTER(C.c_uint32)), shape=(d.numMeshNodes, 8))[mesh.firstNode:mesh.firstNode + mesh.numNodes].copy()
    triangles = np.ctypeslib.as_array(C.cast(d.triangles, C.POINTER(C.c_float)), shape=(d.numTriangles, 9))[mesh.firstTriangle:mesh.firstTriangle + mesh.numTriangles].copy()
    inv = np.array(d.objects[0].invTransform[:], dtype=np.float32)
    assert np.array_equal(inv.reshape(4, 4), np.eye(4, dtype=np.float32)), "the model walks in world space: the mesh object must not be transformed"
    sun = None
    for i in range(d.numLights):
        if d.lights[i].type == 2:
            t = np.array(d.lights[i].transform[:], dtype=np.float32).reshape(4, 4)
            sun = -t[2, :3] if -t[2, 1] > 0 else t[2, :3]
    nodes.tofile(os.path.join(out, "nodes.bin")); triangles.astype(np.float32).tofile(os.path.join(out, "tris.bin"))
    vp = ra.Viewport(w, h, seed=20260928, max_ray_depth=depth)
    rays = []
    for p_index in range(2):
        p = vp.next_pass_params(camera)
        for y in range(stride // 2 + p_index, h, stride):
            for x in range(stride // 2 + 3 * p_index, w, stride):
                v = oracle_lib.render_pixel_paths(scene.desc, p, w, h, x, y)
                for k in range(len(v)):
                    # origin, direction, hit distance, bounce, hit point, shading normal
                    rays.append(np.concatenate([v[k, 0:6], v[k, 8:9], [float(k)], v[k, 11:14], v[k, 14:17]]))
    rays = np.array(rays, dtype=np.float32)
    rays.tofile(os.path.join(out, "rays.bin"))
    with open(os.path.join(out, "meta.txt"), "w") as f:
        f.write("%d %d %d %.9g %.9g %.9g\n" % (len(nodes), len(triangles), len(rays), sun[0], sun[1], sun[2]))
    print("nodes %d, triangles %d, rays %d (%.2f per path), sun %s -> %s" % (len(nodes), len(triangles), len(rays), len(rays) / max(1, np.count_nonzero(rays[:, 7] == 0)), sun, out))


if __name__ == "__main__":
    main()
