# round-3 probes (gpurun -- 'bash tools/r3_probe.sh'): VALU cadence / L1 microbenchmark, serial launch timeline of the driver's bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r3b
mkdir -p $T
tools/microbench/cadence > $T/cadence.txt 2>&1
RTGPU_LANES=1 rocprofv3 --kernel-trace -d $T/prof_serial -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $T/bench_serial_profiled.json 2>/dev/null
db=$(find $T/prof_serial -name '*.db' | head -1)
python tools/rocpd_summary.py $db > $T/kernel_stats_serial.txt
python tools/rocpd_summary.py --timeline $db > $T/timeline_serial.txt
rm -rf $T/prof_serial
python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^$" | grep "passed\|failed\|error\|^\.\|Error\|assert" > $T/pytest_gpu.log
tail -3 $T/pytest_gpu.log
cat $T/cadence.txt | head -16
