#!/bin/bash
cd /root/repo
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"; }
for base in 4 5 6 8; do for ln in 3 4; do
 echo "base $base lanes $ln:  20: $(RTGPU_PASS_BATCH_BASE=$base RTGPU_LANES=$ln b 20) $(RTGPU_PASS_BATCH_BASE=$base RTGPU_LANES=$ln b 20)   64: $(RTGPU_PASS_BATCH_BASE=$base RTGPU_LANES=$ln b 64)   256: $(RTGPU_PASS_BATCH_BASE=$base RTGPU_LANES=$ln b 256)   8: $(RTGPU_PASS_BATCH_BASE=$base RTGPU_LANES=$ln b 8)"
done; done
