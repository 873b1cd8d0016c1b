#!/bin/bash
cd /root/repo
b() { python bench.py --steps $1 --warmup 5 --no-pmc --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; }
for n in 2 4 8; do
 echo "shard 1/$n, 20 steps, ms per step: default $(BENCH_EMULATE_SHARD=$n b 20) $(BENCH_EMULATE_SHARD=$n b 20)  batch5 $(BENCH_EMULATE_SHARD=$n RTGPU_PASS_BATCH=5 b 20)  batch7 $(BENCH_EMULATE_SHARD=$n RTGPU_PASS_BATCH=7 b 20) batch10 $(BENCH_EMULATE_SHARD=$n RTGPU_PASS_BATCH=10 b 20) batch20 $(BENCH_EMULATE_SHARD=$n RTGPU_PASS_BATCH=20 b 20)"
done
echo "full frame 20 steps: $(b 20)"
for n in 2 4 8; do echo "shard 1/$n, 256 steps: default $(BENCH_EMULATE_SHARD=$n b 256) batch10 $(BENCH_EMULATE_SHARD=$n RTGPU_PASS_BATCH=10 b 256)"; done
