# round 5, GPU call 28: bench.py with the warm-up read-back in front of the timed region (host_readback = the steady-state copy): the plumbing tests, the driver's command,
# longer streams, the emulated shards, two gloo ranks on the device
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_bench_plumbing.py -m gpu -x -q 2>&1 | tail -3 | tee $T/pytest_bench_plumbing.txt
python bench.py --steps 20 --warmup 5 > $T/bench_default_steps20.json 2> $T/bench_default_steps20.err
for steps in 64 256; do python bench.py --steps $steps --warmup 5 --no-pmc --cpu-seconds 0 > $T/bench_steps$steps.json 2>/dev/null; done
for n in 2 4 8; do for steps in 20 256; do BENCH_EMULATE_SHARD=$n python bench.py --steps $steps --warmup 5 --no-pmc --no-cpu-baseline > $T/bench_emulated_shard_${n}_steps$steps.json 2>/dev/null; done; done
BENCH_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | grep '^{' > $T/bench_two_gloo_ranks_one_device.json
for f in $T/bench_default_steps20.json $T/bench_steps*.json $T/bench_emulated*.json $T/bench_two*.json; do echo "$f: $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), 'ms/pass', d['host_readback'])" 2>/dev/null)"; done
