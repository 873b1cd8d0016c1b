cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/s5; mkdir -p $T
python -m pytest tests -q -m gpu 2>&1 | grep "passed\|failed\|error\|^FAILED\|^ERROR" > $T/pytest_gpu.log; cat $T/pytest_gpu.log
