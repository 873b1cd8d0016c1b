# round 5, GPU call 5: why the re-trace launches take 8x longer with the directional light turned to sponza.json's orientation [80, 0, 0]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05e
mkdir -p $T
cp raytracer_amd/lib/librtgpu.so /tmp/keep.so
cp variants/librtgpu_r05a.so raytracer_amd/lib/librtgpu.so
for yaw in 0 20 1 5; do
  echo "== light yaw $yaw (r05a library, 24-entry stacks)"
  BENCH_LIGHT_YAW=$yaw python tools/wide_diag.py 2>&1 | tail -1 | cut -c1-700
done 2>&1 | tee $T/yaw_counters.txt
cp variants/librtgpu_vc.so raytracer_amd/lib/librtgpu.so
for yaw in 0 20; do
  echo "== light yaw $yaw (vc library, 16-entry stacks)"
  BENCH_LIGHT_YAW=$yaw python tools/wide_diag.py 2>&1 | tail -1 | cut -c1-700
done 2>&1 | tee -a $T/yaw_counters.txt
BENCH_LIGHT_YAW=0 RTGPU_LANES=1 rocprofv3 --kernel-trace --stats -d $T/prof_serial -o r -- python bench.py --steps 10 --warmup 5 --no-pmc --no-cpu-baseline > $T/bench_serial_profiled.json 2>/dev/null
db=$(find $T/prof_serial -name '*.db' | head -1)
python tools/rocpd_summary.py $db > $T/kernel_stats_serial_yaw0.txt
python tools/rocpd_summary.py --timeline $db 2>/dev/null | head -120 > $T/timeline_serial_yaw0.txt
rm -rf $T/prof_serial
head -12 $T/kernel_stats_serial_yaw0.txt
sed -n 60,110p $T/timeline_serial_yaw0.txt
cp /tmp/keep.so raytracer_amd/lib/librtgpu.so
