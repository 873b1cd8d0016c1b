# What bounds k_trace / k_shade?  Texture-addresser (TA), L1 (TCP) and data-return (TD) counters of `RTGPU_LANES=1 python bench.py`,
# one rocprofv3 --pmc pass per counter group (kernel-trace only).  usage: tools/pmc_diag.sh <tag> [bench args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-diag}; shift
ARGS=${@:---steps 24 --warmup 8}
mkdir -p gpurun_out/$T
i=0
for group in \
  "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  ; do
  i=$((i+1))
  RTGPU_LANES=1 timeout 300 rocprofv3 --kernel-trace --pmc $group -d gpurun_out/$T/p$i -o r -- python bench.py --no-cpu-baseline $ARGS > gpurun_out/$T/bench_p$i.json 2> gpurun_out/$T/err_p$i.txt
  db=$(find gpurun_out/$T/p$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db > gpurun_out/$T/pmc_$i.txt; else echo "pass $i: no db"; tail -3 gpurun_out/$T/err_p$i.txt; fi
  rm -rf gpurun_out/$T/p$i
done
cat gpurun_out/$T/pmc_*.txt | grep -v "^#" | grep "k_trace<24, false>\|k_shade<true" | grep -v " 256$"
