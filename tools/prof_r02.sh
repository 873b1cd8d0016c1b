# end-of-round-2 evidence (run on the GPU box: gpurun -- 'bash tools/prof_r02.sh'); summaries land in gpurun_out/r02/, copied to profiles/r02_*
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r02
mkdir -p $T
# 1. what the driver runs (in-run roofline: HIP events + two --pmc child runs; CPU baseline = the reference's objects)
python bench.py --steps 20 --warmup 5 > $T/bench_default_steps20.json 2> $T/bench_default_steps20.err
# 2. the same passes under rocprofv3 --kernel-trace --stats with serial lanes: the population the roofline's launch time is quoted on
RTGPU_LANES=1 rocprofv3 --kernel-trace --stats -d $T/prof_serial -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $T/bench_serial_profiled.json 2>/dev/null
python tools/rocpd_summary.py $(find $T/prof_serial -name '*.db' | head -1) > $T/kernel_stats_serial.txt
rm -rf $T/prof_serial
# 3. longer streams and the other workloads
for steps in 64 256; do python bench.py --steps $steps --warmup 5 --no-pmc --cpu-seconds 0 > $T/bench_steps$steps.json 2>/dev/null; done
python bench.py --workload bdpt-glass --steps 20 --warmup 5 > $T/bench_config5_bdpt_glass.json 2>/dev/null
python bench.py --workload sponza-textured --steps 64 --warmup 5 --no-pmc --cpu-seconds 0 > $T/bench_sponza_textured.json 2>/dev/null
python bench.py --workload cornell --width 640 --height 480 --depth 4 --steps 16 --warmup 4 --no-pmc --cpu-seconds 0 > $T/bench_config1_cornell.json 2>/dev/null
python bench.py --workload sphere --steps 64 --warmup 5 --no-pmc --cpu-seconds 0 > $T/bench_config2_sphere.json 2>/dev/null
# 4. what bounds k_trace_wide: SQ / TCP / TCC counters, one --pmc pass per group
i=0
for group in \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  ; do
  i=$((i+1))
  RTGPU_LANES=1 timeout 300 rocprofv3 --kernel-trace --pmc $group -d $T/p$i -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > /dev/null 2> $T/err_p$i.txt
  db=$(find $T/p$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db > $T/wide_pmc_$i.txt; else echo "pass $i: no db"; tail -3 $T/err_p$i.txt; fi
  rm -rf $T/p$i $T/err_p$i.txt
done
# 5. the GPU test suite
python -m pytest tests -q -m gpu 2>&1 | grep -v "^$" | grep "passed\|failed\|error\|^\." > $T/pytest_gpu.log
for f in $T/bench_*.json; do echo "$f: $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['unit'], d['config']['workload'][:60])")"; done
head -9 $T/kernel_stats_serial.txt; cat $T/pytest_gpu.log
grep "k_trace_wide" $T/wide_pmc_*.txt | head -40
