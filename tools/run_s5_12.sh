cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/s5; mkdir -p $T
python tools/oracle_fuzz.py 720 606 2>&1 | tail -3 > $T/fuzz_oracle.txt
python tools/wide_fuzz.py 300 88 2>&1 | tail -2 > $T/fuzz_wide.txt
cat $T/fuzz_oracle.txt $T/fuzz_wide.txt
