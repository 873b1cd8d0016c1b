# round 5, GPU call 32: call 31's sweep stopped at RTGPU_TAIL_BLOCKS_PER_CU=5 (the call's limit) -- does a 1/8 shard with that grid finish?  Each run under its own timeout.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05x
mkdir -p $T
for n in 5 2 6; do
  echo "RTGPU_TAIL_BLOCKS_PER_CU=$n:"
  BENCH_EMULATE_SHARD=8 RTGPU_TAIL_BLOCKS_PER_CU=$n timeout 75 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>$T/tail_blocks_$n.err | tail -1 | python -c "
import sys, json
t = sys.stdin.read().strip()
if not t: print('  no JSON line (timeout or error)')
else:
    d = json.loads(t); print('  %.1f Msamples/s %.3f ms/pass' % (d['value'], d['ms_per_step']), d['kernel_time_ms'])"
  echo "  exit status of the pipeline's bench: ${PIPESTATUS[0]}"; tail -2 $T/tail_blocks_$n.err
done 2>&1 | tee $T/tail_blocks_diag.txt
