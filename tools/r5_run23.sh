# round 5, GPU call 23: the mid-walk hand-over test; claim order reversed in k_trace_wide (any-hit requests first, closest-hit rays last: which kind's long rays make the drain?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05w
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mid_walk or wide_traversal" 2>&1 | tail -3 | tee $T/parity_midwalk.txt
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_WIDE_REVERSE=0 RTGPU_WIDE_REVERSE=1 2>&1 | tee $T/ab_reverse.txt
BENCH_EMULATE_SHARD=8 bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_WIDE_REVERSE=0 RTGPU_WIDE_REVERSE=1 2>&1 | tee -a $T/ab_reverse.txt
