# PathTracerMIS evidence set: default bench, serial kernel stats, FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-v15}
mkdir -p gpurun_out/$T
python bench.py > gpurun_out/$T/bench_default.json 2>/dev/null
tail -1 gpurun_out/$T/bench_default.json | cut -c1-400
RTGPU_LANES=1 rocprofv3 --kernel-trace --stats -d gpurun_out/$T/p1 -o r -- python bench.py --no-cpu-baseline > gpurun_out/$T/bench_serial_profiled.json 2>/dev/null
python tools/rocpd_summary.py $(find gpurun_out/$T/p1 -name '*.db' | head -1) > gpurun_out/$T/kernel_stats_serial.txt
head -8 gpurun_out/$T/kernel_stats_serial.txt
for c in FETCH_SIZE WRITE_SIZE; do
  RTGPU_LANES=1 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$T/p_$c -o r -- python bench.py --no-cpu-baseline > /dev/null 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/$T/p_$c -name '*.db' | head -1) > gpurun_out/$T/pmc_$c.txt
  grep -i "k_trace\|k_shade" gpurun_out/$T/pmc_$c.txt | tail -6
done
rm -rf gpurun_out/$T/p1 gpurun_out/$T/p_FETCH_SIZE gpurun_out/$T/p_WRITE_SIZE
