set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/v13
python bench.py > gpurun_out/v13/bench_default.json 2> gpurun_out/v13/bench_default.err
tail -1 gpurun_out/v13/bench_default.json | cut -c1-600
RTGPU_LANES=1 rocprofv3 --kernel-trace --stats -d gpurun_out/v13/prof_serial -o r -- python bench.py > gpurun_out/v13/bench_serial_profiled.json 2>/dev/null
python tools/rocpd_summary.py $(find gpurun_out/v13/prof_serial -name '*.db' | head -1) > gpurun_out/v13/kernel_stats_serial.txt
head -12 gpurun_out/v13/kernel_stats_serial.txt
python tools/bench_vcm.py > gpurun_out/v13/vcm_caustics.json 2>/dev/null; tail -1 gpurun_out/v13/vcm_caustics.json
python tools/bench_vcm.py --scene sponza --passes 16 > gpurun_out/v13/vcm_sponza.json 2>/dev/null; tail -1 gpurun_out/v13/vcm_sponza.json
python tools/bench_vcm.py --scene sponza --passes 16 --no-merging > gpurun_out/v13/vcm_sponza_bdpt.json 2>/dev/null; tail -1 gpurun_out/v13/vcm_sponza_bdpt.json
rocprofv3 --kernel-trace --stats -d gpurun_out/v13/prof_vcm -o r -- python tools/bench_vcm.py --scene sponza > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/v13/prof_vcm -name '*.db' | head -1) > gpurun_out/v13/vcm_kernel_stats_sponza.txt
head -24 gpurun_out/v13/vcm_kernel_stats_sponza.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/v13/prof_vcm2 -o r -- python tools/bench_vcm.py > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/v13/prof_vcm2 -name '*.db' | head -1) > gpurun_out/v13/vcm_kernel_stats_caustics.txt
head -24 gpurun_out/v13/vcm_kernel_stats_caustics.txt
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/v13/pytest_gpu.log; cat gpurun_out/v13/pytest_gpu.log
rm -rf gpurun_out/v13/prof_serial gpurun_out/v13/prof_vcm gpurun_out/v13/prof_vcm2
