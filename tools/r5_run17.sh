# round 5, GPU call 17: re-trace launches with a device-side adaptive grid (full traversal grid when the queues hold more than 1024 requests per CU) -- a delta sun along an axis, the headline, parity
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05q
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "axis or retrace or wide or tail" 2>&1 | tail -3 | tee $T/pytest.log
cat > /tmp/sun.py <<'PY'
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import numpy as np, raytracer_amd as ra
from raytracer_amd import scenes
w, h = 1920, 1080
pos, idx, nrm, tan, uv, mat = scenes.sponza_class_mesh(262144, 7, refine=True)
for orientation in ((0.0, 0.0, 0.0), (90.0, 0.0, 0.0), (80.0, 20.0, 0.0)):
    scene = ra.Scene()
    mats = [scene.add_material("diffuse", c) for _, c in scenes.SPONZA_MATERIALS]
    scene.add_mesh(pos, idx, nrm, tan, uv, mat, mats)
    scene.add_background_light((1.0, 1.5, 2.0))
    scene.add_directional_light((20000.0, 19000.0, 18000.0), 0.0, ra.transform_from_euler((0.0, 0.0, 0.0), orientation))
    scene.build()
    camera = ra.Camera((-12.5, 2.2, 0.6), (4.0, 82.0, 0.0), w / h, 65.0)
    vp = ra.Viewport(w, h, seed=515, max_ray_depth=8); vp.set_renderer(scene, intersection_counters=False)
    ctx = vp.device_context(); lib = ra.rtgpu_lib()
    vp.render(camera, 5); lib.rtgpu_synchronize(ctx); c0 = vp.counters()
    t0 = time.perf_counter(); vp.render(camera, 20); lib.rtgpu_synchronize(ctx); dt = time.perf_counter() - t0
    c1 = vp.counters()
    print("delta sun, orientation", orientation, ": %.1f Msamples/s" % ((c1["numRays"] - c0["numRays"]) / dt / 1e6), "retraced", c1["numRetracedRays"] - c0["numRetracedRays"], "shadow rays", c1["numShadowRays"] - c0["numShadowRays"], flush=True)
PY
for e in RTGPU_RETRACE_FULL_GRID=0 RTGPU_RETRACE_FULL_GRID=1; do echo "== $e"; env $e python /tmp/sun.py 2>/dev/null; done | tee $T/delta_sun_grid.txt
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_RETRACE_FULL_GRID=0 RTGPU_RETRACE_FULL_GRID=1 2>&1 | tee $T/ab_grid.txt
