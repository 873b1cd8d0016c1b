#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "wide" 2>&1 | grep -v "^$" | tail -4
RTGPU_WIDE_DIAG=1 python tools/wide_diag.py 2>&1 | tail -3
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), {k: round(v,1) for k,v in d.get('kernel_time_ms',{}).items()})"; }
for i in 1 2 3; do echo "new 64: $(b 64)"; echo "new block64 64: $(RTGPU_WIDE_BLOCK64=1 b 64)"; done
for i in 1 2; do echo "new 20: $(b 20)   block64 20: $(RTGPU_WIDE_BLOCK64=1 b 20)   new 256: $(b 256)   block64 256: $(RTGPU_WIDE_BLOCK64=1 b 256)"; done
