# round 5, GPU call 22: a traversal launch lasts 0.32 ms however few rays it holds (1.40 ms at 6 passes per launch, 3.26 ms at 17: t = 0.32 + 0.173 n) -- its
# longest rays.  Experiment: a wave whose queue ran dry N loop iterations ago hands what it still walks to the re-trace launch (RTGPU_WIDE_DRAIN_ABORT=N),
# with and without the cooperative walker behind it.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05w
mkdir -p $T
RTGPU_WIDE_DRAIN_ABORT=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sponza or oracle or bit_exact" 2>&1 | tail -3 | tee $T/parity_abort4.txt
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_WIDE_DRAIN_ABORT=0 RTGPU_WIDE_DRAIN_ABORT=64 RTGPU_WIDE_DRAIN_ABORT=32 RTGPU_WIDE_DRAIN_ABORT=16 RTGPU_WIDE_DRAIN_ABORT=8 "RTGPU_WIDE_DRAIN_ABORT=32 RTGPU_RETRACE_MONSTERS=1" "RTGPU_WIDE_DRAIN_ABORT=16 RTGPU_RETRACE_MONSTERS=1" 2>&1 | tee $T/ab_full.txt
export BENCH_EMULATE_SHARD=8
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_WIDE_DRAIN_ABORT=0 "RTGPU_WIDE_DRAIN_ABORT=32 RTGPU_LOCAL_EXACT=0" "RTGPU_WIDE_DRAIN_ABORT=16 RTGPU_LOCAL_EXACT=0" "RTGPU_WIDE_DRAIN_ABORT=0 RTGPU_LOCAL_EXACT=0" "RTGPU_WIDE_DRAIN_ABORT=16 RTGPU_LOCAL_EXACT=0 RTGPU_RETRACE_MONSTERS=1" "RTGPU_WIDE_DRAIN_ABORT=8 RTGPU_LOCAL_EXACT=0 RTGPU_RETRACE_MONSTERS=1" 2>&1 | tee $T/ab_shard8.txt
