set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-v16}
mkdir -p gpurun_out/$T
python -m pytest tests/test_gpu_vcm.py -q 2>&1 | tail -3 > gpurun_out/$T/pytest_vcm.log; cat gpurun_out/$T/pytest_vcm.log
python tools/bench_vcm.py --passes 16 > gpurun_out/$T/vcm_caustics.json 2>/dev/null; tail -1 gpurun_out/$T/vcm_caustics.json
python tools/bench_vcm.py --scene sponza --passes 16 > gpurun_out/$T/vcm_sponza.json 2>/dev/null; tail -1 gpurun_out/$T/vcm_sponza.json
python tools/bench_vcm.py --scene sponza --passes 16 --no-merging > gpurun_out/$T/vcm_sponza_bdpt.json 2>/dev/null; tail -1 gpurun_out/$T/vcm_sponza_bdpt.json
rocprofv3 --kernel-trace --stats -d gpurun_out/$T/prof_vcm -o r -- python tools/bench_vcm.py --scene sponza --passes 16 > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/$T/prof_vcm -name '*.db' | head -1) > gpurun_out/$T/vcm_kernel_stats_sponza.txt
head -30 gpurun_out/$T/vcm_kernel_stats_sponza.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/$T/prof_vcm2 -o r -- python tools/bench_vcm.py --passes 16 > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/$T/prof_vcm2 -name '*.db' | head -1) > gpurun_out/$T/vcm_kernel_stats_caustics.txt
head -30 gpurun_out/$T/vcm_kernel_stats_caustics.txt
rm -rf gpurun_out/$T/prof_vcm gpurun_out/$T/prof_vcm2
