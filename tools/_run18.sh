#!/bin/bash
cd /root/repo
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"; }
for i in 1 2 3; do
echo "exact 256: $(b 256)   wide 256: $(RTGPU_WIDE=1 RTGPU_WIDE_EAGER_LEAVES=1 b 256)   wide 256 lanes4: $(RTGPU_WIDE=1 RTGPU_WIDE_EAGER_LEAVES=1 RTGPU_LANES=4 b 256)  exact lanes4: $(RTGPU_LANES=4 b 256)"
echo "exact 20: $(b 20)   wide 20: $(RTGPU_WIDE=1 RTGPU_WIDE_EAGER_LEAVES=1 b 20)   wide 20 lanes4: $(RTGPU_WIDE=1 RTGPU_WIDE_EAGER_LEAVES=1 RTGPU_LANES=4 b 20)"
done
