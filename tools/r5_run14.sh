# round 5, GPU call 14: any-hit subtree sharing earlier in the re-trace launches (RTGPU_RETRACE_SPLIT_AFTER drain iterations instead of 32)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05n
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "wide or packet or full_size or retrace or tail or axis" 2>&1 | tail -3 | tee $T/pytest.log
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_RETRACE_SPLIT_AFTER=32 RTGPU_RETRACE_SPLIT_AFTER=8 RTGPU_RETRACE_SPLIT_AFTER=4 RTGPU_RETRACE_SPLIT_AFTER=2 RTGPU_RETRACE_SPLIT_AFTER=1 2>&1 | tee $T/ab_split.txt
BENCH_EMULATE_SHARD=8 bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_RETRACE_SPLIT_AFTER=32 RTGPU_RETRACE_SPLIT_AFTER=4 2>&1 | tee -a $T/ab_split.txt
