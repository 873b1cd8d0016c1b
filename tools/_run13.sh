#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "wide" -s 2>&1 | tail -15
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), {k: round(v,1) for k,v in d.get('kernel_time_ms',{}).items()})"; }
for i in 1 2 3; do echo "exact 64:"; b 64; echo "wide 64:"; RTGPU_WIDE=1 b 64; done
for i in 1 2 3; do echo "exact 20:"; b 20; echo "wide 20:"; RTGPU_WIDE=1 b 20; done
for i in 1 2 3; do echo "batch8 20:"; RTGPU_PASS_BATCH=8 b 20; echo "batch7 20:"; RTGPU_PASS_BATCH=7 b 20; done
