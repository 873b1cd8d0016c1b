cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/s5; mkdir -p $T
python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -q -x -m gpu 2>&1 | grep "passed\|failed\|error\|^FAILED\|^ERROR\|assert" | head -20 > $T/pytest_fp16.log; cat $T/pytest_fp16.log
bash tools/ab_libs.sh "--steps 64 --warmup 5" base u16 > $T/ab_fp16_64.txt; cat $T/ab_fp16_64.txt
bash tools/ab_libs.sh "--steps 20 --warmup 5" base u16 > $T/ab_fp16_20.txt; cat $T/ab_fp16_20.txt
