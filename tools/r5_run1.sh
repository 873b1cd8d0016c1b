# round 5, GPU call 1: suite on the new library, A/B against round 4's library, walk diagnostics, serial kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r05a
mkdir -p $T
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $T/pytest_gpu.log
tail -3 $T/pytest_gpu.log
bash tools/ab_libs.sh "--steps 20 --warmup 5" r04 base 2>&1 | tee $T/ab_r04_base.txt
bash tools/ab_libs.sh "--steps 64 --warmup 5" r04 base 2>&1 | tee -a $T/ab_r04_base.txt
for e in 1 RTGPU_RETRACE_MONSTERS=0 RTGPU_ABORT_RETRACE_AFTER=32 RTGPU_ABORT_RETRACE_AFTER=256; do :; done
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_RETRACE_MONSTERS=0 RTGPU_ABORT_RETRACE_AFTER=32 RTGPU_ABORT_RETRACE_AFTER=96 RTGPU_ABORT_RETRACE_AFTER=256 2>&1 | tee $T/ab_abort_after.txt
BENCH_EMULATE_SHARD=8 bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_RETRACE_MONSTERS=0 RTGPU_RETRACE_MONSTERS=1 2>&1 | tee $T/ab_monsters_shard8.txt
RTGPU_WIDE_DIAG=1 python tools/wide_diag.py > $T/wide_diag_1.txt 2>&1
RTGPU_WIDE_DIAG=2 python tools/wide_diag.py > $T/wide_diag_2.txt 2>&1
cat $T/wide_diag_1.txt $T/wide_diag_2.txt
RTGPU_LANES=1 rocprofv3 --kernel-trace --stats -d $T/prof_serial -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > $T/bench_serial_profiled.json 2>/dev/null
db=$(find $T/prof_serial -name '*.db' | head -1)
python tools/rocpd_summary.py $db > $T/kernel_stats_serial.txt
python tools/rocpd_summary.py --timeline $db | head -150 > $T/timeline_serial.txt
rm -rf $T/prof_serial
head -14 $T/kernel_stats_serial.txt
