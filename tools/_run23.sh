#!/bin/bash
cd /root/repo
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for n in 4 8; do
 echo "shard 1/$n 20 steps: default $(BENCH_EMULATE_SHARD=$n b 20) $(BENCH_EMULATE_SHARD=$n b 20) | exact $(BENCH_EMULATE_SHARD=$n RTGPU_WIDE=0 b 20) $(BENCH_EMULATE_SHARD=$n RTGPU_WIDE=0 b 20) | lanes6 $(BENCH_EMULATE_SHARD=$n RTGPU_LANES=6 b 20) | batch10 $(BENCH_EMULATE_SHARD=$n RTGPU_PASS_BATCH=10 b 20) $(BENCH_EMULATE_SHARD=$n RTGPU_PASS_BATCH=10 b 20) | batch10 exact $(BENCH_EMULATE_SHARD=$n RTGPU_PASS_BATCH=10 RTGPU_WIDE=0 b 20) | batch10 lanes2 $(BENCH_EMULATE_SHARD=$n RTGPU_PASS_BATCH=10 RTGPU_LANES=2 b 20)"
done
RTGPU_LANES=1 BENCH_EMULATE_SHARD=8 rocprofv3 --kernel-trace --stats -d /tmp/prof8 -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --cpu-seconds 0 > /dev/null 2>&1
python tools/rocpd_summary.py $(find /tmp/prof8 -name '*.db' | head -1) | head -14
