cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp raytracer_amd/lib/librtgpu.so /tmp/keep.so
B="python bench.py --no-cpu-baseline --no-pmc --steps 64 --warmup 8"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), d["kernel_time_ms"])'
for rep in 1 2; do
for cfg in "q5 5" "q6 6" "q6 5"; do
  set -- $cfg
  cp ab/librtgpu_$1.so raytracer_amd/lib/librtgpu.so
  echo -n "$1 blocks/CU=$2: "; RTGPU_TRAV_BLOCKS_PER_CU=$2 $B 2>/dev/null | tail -1 | python -c "$P"
done; done
mkdir -p gpurun_out/r02_quant0
for v in q5 q6; do
cp ab/librtgpu_$v.so raytracer_amd/lib/librtgpu.so
i=0
for group in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  BPC=5; [ $v = q6 ] && BPC=6
  RTGPU_TRAV_BLOCKS_PER_CU=$BPC RTGPU_LANES=1 timeout 300 rocprofv3 --kernel-trace --pmc $group -d gpurun_out/r02_quant0/p$i -o r -- python bench.py --no-cpu-baseline --no-pmc --steps 24 --warmup 8 > /dev/null 2> gpurun_out/r02_quant0/err_${v}_p$i.txt
  db=$(find gpurun_out/r02_quant0/p$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db > gpurun_out/r02_quant0/pmc_${v}_$i.txt; fi
  rm -rf gpurun_out/r02_quant0/p$i
done
done
cp /tmp/keep.so raytracer_amd/lib/librtgpu.so
grep -h "k_trace_quant" gpurun_out/r02_quant0/pmc_q5_*.txt | cut -c1-120
echo ====
grep -h "k_trace_quant" gpurun_out/r02_quant0/pmc_q6_*.txt | cut -c1-120
