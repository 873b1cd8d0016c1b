# round 5, GPU call 27: what the frame's read-back costs -- first call (registers the bitmaps as page-locked memory) against later calls; behind a device sync against queued behind the passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05x
mkdir -p $T
cat > /tmp/readback.py <<'PY'
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
vp.render(camera, 5); lib.rtgpu_synchronize(ctx)
def timed(label, f):
    t0 = time.perf_counter(); f(); dt = time.perf_counter() - t0
    print("%-70s %8.3f ms" % (label, dt * 1e3), flush=True); return dt
timed("first read-back (registers the bitmap), device idle", lambda: host.rth_viewport_fetch_sum(vp._h))
for i in range(3):
    vp.render(camera, 1); lib.rtgpu_synchronize(ctx)
    timed("later read-back, device idle (behind a sync)", lambda: host.rth_viewport_fetch_sum(vp._h))
for i in range(3):
    a = timed("20 passes + sync", lambda: (vp.render(camera, 20), lib.rtgpu_synchronize(ctx)))
    b = timed("  read-back behind it", lambda: host.rth_viewport_fetch_sum(vp._h))
    c = timed("20 passes + read-back queued behind them (no sync between)", lambda: (vp.render(camera, 20), host.rth_viewport_fetch_sum(vp._h)))
    print("  -> read-back inside the region costs %.3f ms, behind a sync %.3f ms" % ((c - a) * 1e3, b * 1e3), flush=True)
PY
python /tmp/readback.py 2>/dev/null | tee $T/readback_cost.txt
