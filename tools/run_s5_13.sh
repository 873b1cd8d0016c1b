cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/s5; mkdir -p $T
python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | grep "passed\|failed\|error\|^FAILED\|^ERROR\|assert\|Error" | head -10
bash tools/ab_env.sh "--steps 64 --warmup 5" RTGPU_PACKET=0 RTGPU_PACKET=1 | cut -c1-250
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_PACKET=0 RTGPU_PACKET=1 | cut -c1-250
RTGPU_LANES=1 rocprofv3 --kernel-trace --stats -d $T/prof_pk -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > /dev/null 2>&1
db=$(find $T/prof_pk -name '*.db' | head -1); python tools/rocpd_summary.py $db | grep "packet\|kernel " | cut -c1-150; rm -rf $T/prof_pk
