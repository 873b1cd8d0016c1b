#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_images.py -q -x -k "wide or reference" 2>&1 | grep -v "^$" | tail -4
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), {k: round(v,1) for k,v in d.get('kernel_time_ms',{}).items()})"; }
for i in 1 2 3; do echo "64: $(b 64)"; done
echo "20: $(b 20)"; echo "256: $(b 256)"
