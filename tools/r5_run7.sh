# round 5, GPU call 7: the whole suite on the round's library, the driver's command with the re-based roofline, N > 1 plumbing, emulated shards
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05g
mkdir -p $T
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $T/pytest_gpu.log
tail -4 $T/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $T/bench_default_steps20.json 2> $T/bench_default_steps20.err
tail -1 $T/bench_default_steps20.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('value', round(d['value'], 1), 'ms/pass', round(d['ms_per_step'], 3))
print({k: r.get(k) for k in ('kernel', 'bound', 'avg_launch_ms', 'launches', 'algorithmic_bytes_per_launch', 'achieved', 'frac', 'traffic', 'traffic_GBs', 'traffic_frac', 'traffic_over_algorithmic', 'traffic_fetch_size_bytes', 'traffic_write_size_bytes')})
print(r.get('algorithmic_model')); print(r.get('ceilings', {}).get('fracs')); print(d.get('kernel_time_ms')); print(d.get('cpu_baseline'))"
tail -3 $T/bench_default_steps20.err
BENCH_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2> $T/two_ranks.err | grep '^{' > $T/bench_two_gloo_ranks_one_device.json
grep "\[bench\]" $T/two_ranks.err
python -c "
import json; d = json.loads(open('$T/bench_two_gloo_ranks_one_device.json').read()); print(round(d['value'], 1), d['multi_gpu'])"
for n in 2 4 8; do BENCH_EMULATE_SHARD=$n python bench.py --steps 20 --warmup 5 --no-pmc --cpu-seconds 0 --no-cpu-baseline > $T/bench_emulated_shard_${n}_steps20.json 2>/dev/null; tail -1 $T/bench_emulated_shard_${n}_steps20.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('shard $n', round(d['value'], 1), round(d['ms_per_step'], 3), d['metric'][-60:], d['kernel_time_ms'])"; done
