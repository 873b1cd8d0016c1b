# round 5, GPU call 33: the fused tail's grid on shards (3 against 4 blocks per CU: call 31 saw +1.5 % once); python bench.py without flags (256 passes) under the timed region's new end
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05x
mkdir -p $T
BENCH_EMULATE_SHARD=8 bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_TAIL_BLOCKS_PER_CU=4 RTGPU_TAIL_BLOCKS_PER_CU=3 RTGPU_TAIL_BLOCKS_PER_CU=4 RTGPU_TAIL_BLOCKS_PER_CU=3 2>&1 | tee $T/ab_tail_grid.txt
BENCH_EMULATE_SHARD=4 bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_TAIL_BLOCKS_PER_CU=4 RTGPU_TAIL_BLOCKS_PER_CU=3 2>&1 | tee -a $T/ab_tail_grid.txt
timeout 400 python bench.py > $T/bench_noflags_256.json 2>/dev/null
tail -1 $T/bench_noflags_256.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print(round(d['value'], 1), d['steps'], round(d['ms_per_step'], 3), r['avg_launch_ms'], r['frac'], r['traffic_frac'], r.get('traffic_over_compulsory'), d['host_readback'])"
