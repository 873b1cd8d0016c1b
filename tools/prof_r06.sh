# end-of-round-6 evidence (run on the GPU box: gpurun -- 'bash tools/prof_r06.sh'); summaries land in gpurun_out/r06/, copied to profiles/r06_*
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r06
mkdir -p $T
# 1. what the driver runs (in-run roofline: HIP events + three --pmc child runs; CPU baseline = the reference's objects)
python bench.py --steps 20 --warmup 5 > $T/bench_default_steps20.json 2> $T/bench_default_steps20.err
# 2. the same passes under rocprofv3 --kernel-trace --stats with serial lanes: the population the roofline's launch time is quoted on; + its timeline
RTGPU_LANES=1 rocprofv3 --kernel-trace --stats -d $T/prof_serial -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $T/bench_serial_profiled.json 2>/dev/null
db=$(find $T/prof_serial -name '*.db' | head -1)
python tools/rocpd_summary.py $db > $T/kernel_stats_serial.txt
python tools/rocpd_summary.py --timeline $db | head -150 > $T/timeline_serial.txt
rm -rf $T/prof_serial
# 2b. the same for a 1/8 shard (what rank 0 of 8 runs): block-local re-trace + the fused tail in the timeline
BENCH_EMULATE_SHARD=8 RTGPU_LANES=1 rocprofv3 --kernel-trace --stats -d $T/prof_shard8 -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > /dev/null 2>&1
db=$(find $T/prof_shard8 -name '*.db' | head -1)
python tools/rocpd_summary.py $db > $T/kernel_stats_serial_shard8.txt
python tools/rocpd_summary.py --timeline $db | tail -80 > $T/timeline_serial_shard8.txt
rm -rf $T/prof_shard8
# 3. the counters behind roofline.traffic and roofline.ceilings, one --pmc pass per group (the same groups bench.py's child runs collect)
i=0
for group in \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  ; do
  i=$((i+1))
  RTGPU_LANES=1 timeout 300 rocprofv3 --kernel-trace --pmc $group -d $T/p$i -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > /dev/null 2> $T/err_p$i.txt
  db=$(find $T/p$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db > $T/pmc_$i.txt; else echo "pass $i: no db"; tail -3 $T/err_p$i.txt; fi
  rm -rf $T/p$i $T/err_p$i.txt
done
# 4. longer streams and the other workloads (CPU baseline: the reference's objects on every PathTracerMIS workload)
for steps in 64 256; do python bench.py --steps $steps --warmup 5 --no-pmc --cpu-seconds 0 > $T/bench_steps$steps.json 2>/dev/null; done
python bench.py --workload sponza-all --steps 20 --warmup 5 --no-pmc > $T/bench_sponza_all.json 2>/dev/null
python bench.py --workload bdpt-glass --steps 20 --warmup 5 --cpu-seconds 0 > $T/bench_config5_bdpt_glass.json 2>/dev/null
python bench.py --workload sponza-textured --steps 64 --warmup 5 --no-pmc > $T/bench_sponza_textured.json 2>/dev/null
python bench.py --workload cornell --width 640 --height 480 --depth 4 --steps 16 --warmup 4 --no-pmc --cpu-seconds 0 > $T/bench_config1_cornell.json 2>/dev/null
python bench.py --workload cornell --steps 32 --warmup 4 --no-pmc --cpu-seconds 0 > $T/bench_cornell_1080p.json 2>/dev/null
python bench.py --workload zoo --steps 32 --warmup 4 --no-pmc --cpu-seconds 0 > $T/bench_zoo_1080p.json 2>/dev/null
python bench.py --workload sphere --steps 64 --warmup 5 --no-pmc --cpu-seconds 0 > $T/bench_config2_sphere.json 2>/dev/null
# 5. what rank 0 of N pays (the tiles it would own; every rank does the same amount of work in parallel): at the driver's step count and at the configuration's 256
for n in 2 4 8; do for steps in 20 256; do BENCH_EMULATE_SHARD=$n python bench.py --steps $steps --warmup 5 --no-pmc --no-cpu-baseline > $T/bench_emulated_shard_${n}_steps$steps.json 2>/dev/null; done; done
# 5b. bench.py's N > 1 path end to end on this box: two gloo ranks sharing the device (per-rank rays / times, gather time, one-GPU frame check)
BENCH_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | grep '^{' > $T/bench_two_gloo_ranks_one_device.json
# 5c. the memory knobs INTEGRATION.md section 3 quotes (a co-tenant's levers; results do not depend on them)
for E in "RTGPU_LANES=1" "RTGPU_LANE_BUDGET_MB=4096" "RTGPU_MAX_STREAM_BATCH=8"; do for steps in 20 256; do
  env $E python bench.py --steps $steps --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-28s steps %3d  %8.1f Msamples/s  %.3f ms/pass' % ('$E', $steps, d['value'], d['ms_per_step']))" >> $T/memory_knobs.txt
done; done
# 5d. how the batch lanes overlap in the driver's timed region (tools/prof_concurrency.sh)
bash tools/prof_concurrency.sh > /dev/null 2>&1; cp gpurun_out/probes/concurrency_all.txt $T/concurrency.txt 2>/dev/null
# 6. the GPU test suite and the smoke test
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$" | grep "passed\|failed\|error\|^\." > $T/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "^smoke" > $T/smoke.log
for f in $T/bench_*.json; do echo "$f: $(tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms/pass', d['config']['workload'][:70])" 2>/dev/null)"; done
head -12 $T/kernel_stats_serial.txt; cat $T/pytest_gpu.log $T/smoke.log
