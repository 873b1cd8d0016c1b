# soak runs on the GPU box: device against the CPU oracle, 4-wide / packet walks against the binary walk (gpurun -- 'bash tools/fuzz_gpu.sh [oracle seconds] [seed] [wide seconds] [seed]'; profiles/r04_fuzz.txt)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/probes; mkdir -p $T
python tools/oracle_fuzz.py ${1:-720} ${2:-606} 2>&1 | tail -3 > $T/fuzz_oracle.txt
python tools/wide_fuzz.py ${3:-300} ${4:-88} 2>&1 | tail -2 > $T/fuzz_wide.txt
cat $T/fuzz_oracle.txt $T/fuzz_wide.txt
