#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "wide" -s 2>&1 | tail -6
python - <<'PY'
import os
os.environ["RTGPU_WIDE"]="1"
import raytracer_amd as ra
from raytracer_amd import scenes
w,h=1920,1080
scene,camera=scenes.sponza_class(w/h)
vp=ra.Viewport(w,h,seed=515,max_ray_depth=8); vp.set_renderer(scene)
ra.rtgpu_lib().rtgpu_set_intersection_counters(vp.device_context(),0)
vp.render(camera,4); c=vp.counters()
print({k:c[k] for k in ("numRays","numShadowRays","numRetracedRays","numUntrustedRays","numStackOverflowRays")})
PY
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), {k: round(v,1) for k,v in d.get('kernel_time_ms',{}).items()})"; }
for i in 1 2; do echo "exact 64:"; b 64; echo "wide 64:"; RTGPU_WIDE=1 b 64; done
for o in 16 24 40 48; do echo "wide other=$o:"; RTGPU_WIDE=1 RTGPU_OTHER_MIN_LANES=$o b 64; done
for o in 16 40 52; do echo "wide refill=$o:"; RTGPU_WIDE=1 RTGPU_REFILL_MIN_IDLE=$o b 64; done
for o in 4 6; do echo "wide blocks=$o:"; RTGPU_WIDE=1 RTGPU_TRAV_BLOCKS_PER_CU=$o b 64; done
