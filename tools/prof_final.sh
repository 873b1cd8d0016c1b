# end-of-round evidence: default bench, the same command under rocprofv3 (serial lanes), GPU tests, VCM throughput + kernel stats
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-v18}
mkdir -p gpurun_out/$T
python bench.py > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err
tail -1 gpurun_out/$T/bench_default.json | cut -c1-400
RTGPU_LANES=1 rocprofv3 --kernel-trace --stats -d gpurun_out/$T/prof_serial -o r -- python bench.py --no-cpu-baseline > gpurun_out/$T/bench_serial_profiled.json 2>/dev/null
python tools/rocpd_summary.py $(find gpurun_out/$T/prof_serial -name '*.db' | head -1) > gpurun_out/$T/kernel_stats_serial.txt
head -8 gpurun_out/$T/kernel_stats_serial.txt
python -c "
import json
d=json.loads(open('gpurun_out/$T/bench_serial_profiled.json').read().strip().splitlines()[-1])
print('HIP events avg_launch_ms', d['roofline']['avg_launch_ms'], 'launches', d['roofline']['launches'])"
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/$T/pytest_gpu.log; cat gpurun_out/$T/pytest_gpu.log
python tools/bench_vcm.py --passes 16 > gpurun_out/$T/vcm_caustics.json 2>/dev/null; tail -1 gpurun_out/$T/vcm_caustics.json | cut -c1-300
python tools/bench_vcm.py --passes 16 --no-merging > gpurun_out/$T/vcm_caustics_bdpt.json 2>/dev/null; tail -1 gpurun_out/$T/vcm_caustics_bdpt.json | cut -c1-300
python tools/bench_vcm.py --scene sponza --passes 16 > gpurun_out/$T/vcm_sponza.json 2>/dev/null; tail -1 gpurun_out/$T/vcm_sponza.json | cut -c1-300
python tools/bench_vcm.py --scene sponza --passes 16 --no-merging > gpurun_out/$T/vcm_sponza_bdpt.json 2>/dev/null; tail -1 gpurun_out/$T/vcm_sponza_bdpt.json | cut -c1-300
rocprofv3 --kernel-trace --stats -d gpurun_out/$T/prof_vcm -o r -- python tools/bench_vcm.py --passes 16 > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/$T/prof_vcm -name '*.db' | head -1) > gpurun_out/$T/vcm_kernel_stats_caustics.txt
head -9 gpurun_out/$T/vcm_kernel_stats_caustics.txt
rm -rf gpurun_out/$T/prof_serial gpurun_out/$T/prof_vcm
