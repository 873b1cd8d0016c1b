# FETCH_SIZE / WRITE_SIZE of `RTGPU_LANES=1 python bench.py` (separate runs, kernel-trace only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-v19}
mkdir -p gpurun_out/$T
for c in FETCH_SIZE WRITE_SIZE; do
  RTGPU_LANES=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$T/p_$c -o r -- python bench.py --no-cpu-baseline > /dev/null 2>&1
  python tools/rocpd_summary.py $(find gpurun_out/$T/p_$c -name '*.db' | head -1) > gpurun_out/$T/pmc_$c.txt
  grep -i "k_trace\|k_shade" gpurun_out/$T/pmc_$c.txt | tail -3
  rm -rf gpurun_out/$T/p_$c
done
