#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "wide" -s 2>&1 | grep -v "^$" | tail -6
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), {k: round(v,1) for k,v in d.get('kernel_time_ms',{}).items()})"; }
for i in 1 2 3; do echo "lazy gate 64: $(b 64)"; echo "eager gate 64: $(RTGPU_WIDE_EAGER_GATE=1 b 64)"; done
