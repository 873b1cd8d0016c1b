# round 5, GPU call 29: where the host is during the driver's timed region -- time stamps of the 20 Render() calls and of the final synchronisation (no profiler: its launch overhead would be the answer)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05x
mkdir -p $T
cat > /tmp/submit.py <<'PY'
import sys, time, os, ctypes as C
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, raytracer_amd as ra
import bench
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5"]
args = bench.parse_args()
w, h = args.width, args.height
scene, camera = bench.build_scene(args, w / h)
vp = bench.make_viewport(ra, args, scene, 0, None)
lib = ra.rtgpu_lib(); host = ra.host_lib(); ctx = vp.device_context()
lib.rtgpu_set_intersection_counters(ctx, 0)
vp.render(camera, 5); host.rth_viewport_fetch_sum(vp._h); lib.rtgpu_synchronize(ctx); vp.counters()
for rep in range(3):
    lib.rtgpu_synchronize(ctx)
    t0 = time.perf_counter(); stamps = []
    for i in range(20):
        vp.render(camera, 1); stamps.append((time.perf_counter() - t0) * 1e3)
    lib.rtgpu_synchronize(ctx); total = (time.perf_counter() - t0) * 1e3
    print("rep %d: Render() returned at (ms)" % rep, " ".join("%.2f" % s for s in stamps), "| synchronised at %.2f ms" % total, flush=True)
for rep in range(2):
    lib.rtgpu_synchronize(ctx)
    t0 = time.perf_counter(); vp.render(camera, 20); t1 = (time.perf_counter() - t0) * 1e3
    lib.rtgpu_synchronize(ctx); total = (time.perf_counter() - t0) * 1e3
    print("one call of 20 passes returned at %.2f ms | synchronised at %.2f ms" % (t1, total), flush=True)
PY
python /tmp/submit.py 2>/dev/null | tee $T/host_submit_stamps.txt
