#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
S=$SECONDS
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r02_gpu_suite.log
echo "suite wall $((SECONDS-S)) s" >> gpurun_out/r02_gpu_suite.log
cat gpurun_out/r02_gpu_suite.log
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"; }
for i in 1 2 3; do
echo "default 20: $(b 20)  lanes4 20: $(RTGPU_LANES=4 b 20)  exact 20: $(RTGPU_WIDE=0 b 20)   default 64: $(b 64)  lanes4 64: $(RTGPU_LANES=4 b 64)  default 256: $(b 256)"
done
