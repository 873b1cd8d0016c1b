# round 5, last GPU call: the suite, the driver's command and the delta-sun check on the final library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05final
mkdir -p $T
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$" | grep "passed\|failed\|error\|^\." > $T/pytest_gpu.log
tail -2 $T/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $T/bench_default_steps20.json 2> $T/bench_default_steps20.err
tail -1 $T/bench_default_steps20.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print(round(d['value'], 1), round(d['ms_per_step'], 3), r['avg_launch_ms'], r['frac'], r['traffic_frac'], d['kernel_time_ms'])"
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
python /tmp/sun.py 2>/dev/null | tee $T/delta_sun_default.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "^smoke" | tee $T/smoke.log
