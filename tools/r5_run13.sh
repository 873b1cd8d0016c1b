# round 5, GPU call 13: node layout probe (a node shares its 128-byte line with its largest interior child instead of a sibling); the re-trace's hand-over threshold
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05m
mkdir -p $T
RTGPU_WIDE_LAYOUT=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wide or packet or full_size or axis" 2>&1 | tail -3 | tee $T/pytest_layout.log
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_WIDE_LAYOUT=0 RTGPU_WIDE_LAYOUT=1 RTGPU_ABORT_RETRACE_AFTER=32 2>&1 | tee $T/ab_layout.txt
bash tools/ab_env.sh "--steps 64 --warmup 5" RTGPU_WIDE_LAYOUT=0 RTGPU_WIDE_LAYOUT=1 2>&1 | tee -a $T/ab_layout.txt
