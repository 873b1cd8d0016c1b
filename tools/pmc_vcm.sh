cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/v17
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY -d gpurun_out/v17/p_sq -o r -- python tools/bench_vcm.py --passes 8 > /dev/null 2>&1
python tools/rocpd_summary.py $(find gpurun_out/v17/p_sq -name '*.db' | head -1) > gpurun_out/v17/vcm_pmc_sq_caustics.txt
grep -i "k_vcm_merge\|k_vcm_camera_shade\|k_vcm_connect" gpurun_out/v17/vcm_pmc_sq_caustics.txt
# (a FETCH_SIZE / WRITE_SIZE pass over this command did not finish within 15 minutes on the pool: the hipCUB sort launches hundreds of tiny kernels per pass)
rm -rf gpurun_out/v17/p_sq
