#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "wide" 2>&1 | tail -4
echo "--- wide postponed"; RTGPU_WIDE=1 RTGPU_WIDE_DIAG=1 python tools/wide_diag.py 2>&1 | tail -3
echo "--- wide eager"; RTGPU_WIDE=1 RTGPU_WIDE_DIAG=1 RTGPU_WIDE_EAGER_LEAVES=1 python tools/wide_diag.py 2>&1 | tail -3
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), {k: round(v,1) for k,v in d.get('kernel_time_ms',{}).items()})"; }
for i in 1 2; do echo "wide postponed 64:"; RTGPU_WIDE=1 b 64; echo "wide eager 64:"; RTGPU_WIDE=1 RTGPU_WIDE_EAGER_LEAVES=1 b 64; done
for o in 16 24 40; do echo "wide postponed other=$o:"; RTGPU_WIDE=1 RTGPU_OTHER_MIN_LANES=$o b 64; done
