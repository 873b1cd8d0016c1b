cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/final; mkdir -p $T
python -m pytest tests -q -m gpu 2>&1 | grep "passed\|failed\|error\|^FAILED\|^ERROR" > $T/pytest_gpu.log; cat $T/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "^smoke" > $T/smoke.log; cat $T/smoke.log
python tools/oracle_fuzz.py 600 707 2>&1 | tail -1 > $T/fuzz_oracle.txt; cat $T/fuzz_oracle.txt
python tools/wide_fuzz.py 240 99 2>&1 | tail -1 > $T/fuzz_wide.txt; cat $T/fuzz_wide.txt
python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
