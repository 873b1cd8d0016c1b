#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multi_device.py -q -x 2>&1 | tail -5
for pb in 4 5 7 8 10 12 20 24; do for ln in 2 3 4; do
  v=$(RTGPU_PASS_BATCH=$pb RTGPU_LANES=$ln python bench.py --steps 20 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
  echo "batch $pb lanes $ln : $v"
done; done | tee gpurun_out/r02_batch_sweep20.txt
echo default; python bench.py --steps 20 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))"
