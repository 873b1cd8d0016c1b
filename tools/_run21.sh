#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
S=$SECONDS
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r02_gpu_suite.log
echo "suite wall $((SECONDS-S)) s" >> gpurun_out/r02_gpu_suite.log
cat gpurun_out/r02_gpu_suite.log
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"; }
for i in 1 2 3; do echo "20: $(b 20)   64: $(b 64)  256: $(b 256)"; done
S=$SECONDS
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_driverlike.json 2> gpurun_out/bench_r02_err.txt
echo "driver-like wall $((SECONDS-S)) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r02_driverlike.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d["roofline"], indent=1)); print(d.get("kernel_time_ms")); print(d.get("cpu_baseline"))
PY
tail -3 gpurun_out/bench_r02_err.txt
