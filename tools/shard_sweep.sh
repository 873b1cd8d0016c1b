cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-60s %8.1f Msamples/s %7.3f ms/pass' % ('$1', d['value'], d['ms_per_step']))
"; }
for rep in 1 2; do
for e in "BENCH_EMULATE_SHARD=8" "BENCH_EMULATE_SHARD=8 RTGPU_PASS_BATCH=5" "BENCH_EMULATE_SHARD=8 RTGPU_PASS_BATCH=7" "BENCH_EMULATE_SHARD=8 RTGPU_PASS_BATCH=20" "BENCH_EMULATE_SHARD=8 RTGPU_PASS_BATCH=10 RTGPU_LANES=2" "BENCH_EMULATE_SHARD=4" "BENCH_EMULATE_SHARD=4 RTGPU_PASS_BATCH=5" "BENCH_EMULATE_SHARD=4 RTGPU_PASS_BATCH=7" "BENCH_EMULATE_SHARD=4 RTGPU_PASS_BATCH=10"; do run "$e"; done
done
