cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/s5; mkdir -p $T
python bench.py --workload bdpt-glass --steps 20 --warmup 5 --cpu-seconds 0 > $T/bench_config5_bdpt_glass.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $T/prof_bdpt -o r -- python bench.py --workload bdpt-glass --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > /dev/null 2>&1
db=$(find $T/prof_bdpt -name '*.db' | head -1); python tools/rocpd_summary.py $db > $T/kernel_stats_config5_bdpt_glass.txt; rm -rf $T/prof_bdpt
tail -1 $T/bench_config5_bdpt_glass.json | cut -c1-400; head -14 $T/kernel_stats_config5_bdpt_glass.txt | cut -c1-150
