# round 5, GPU call 11: axis-parallel any-hit rays inside the 4-wide walk -- parity (new tests), A/B against the previous library, a delta sun straight down
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05k
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -q -x 2>&1 | tail -5 > $T/pytest_parity.log
tail -3 $T/pytest_parity.log
bash tools/ab_libs.sh "--steps 20 --warmup 5" r05m base 2>&1 | tee $T/ab_degenerate_in_walk.txt
cat > /tmp/sun.py <<'PY'
import sys, time, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import numpy as np, raytracer_amd as ra
from raytracer_amd import scenes
w, h = 1920, 1080
pos, idx, nrm, tan, uv, mat = scenes.sponza_class_mesh(262144, 7, refine=True)
for orientation in ((0.0, 0.0, 0.0), (80.0, 20.0, 0.0)):
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
for lib in r05m base; do
  if [ $lib = base ]; then cp /tmp/librtgpu_base.so raytracer_amd/lib/librtgpu.so 2>/dev/null; else cp raytracer_amd/lib/librtgpu.so /tmp/librtgpu_base.so; cp variants/librtgpu_$lib.so raytracer_amd/lib/librtgpu.so; fi
  echo "== library $lib"; python /tmp/sun.py 2>/dev/null
done 2>&1 | tee $T/delta_sun.txt
cp /tmp/librtgpu_base.so raytracer_amd/lib/librtgpu.so
