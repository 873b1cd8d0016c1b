# round 5, GPU call 8: k_trace_wide at six waves per SIMD (80 VGPRs, spills only in the leaf phase); config 5's bimodality probe
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05h
mkdir -p $T
cp raytracer_amd/lib/librtgpu.so /tmp/librtgpu_base.so
for rep in 1 2; do
for cfg in "base 5" "w6 5" "w6 6" "w6 7"; do
  set -- $cfg
  if [ $1 = base ]; then cp /tmp/librtgpu_base.so raytracer_amd/lib/librtgpu.so; else cp variants/librtgpu_$1.so raytracer_amd/lib/librtgpu.so; fi
  touch raytracer_amd/lib/librtgpu.so raytracer_amd/lib/libraytracer_amd_host.so raytracer_amd/lib/rt_demo
  RTGPU_TRAV_BLOCKS_PER_CU=$2 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$1 blocks/CU $2: %.1f Msamples/s, %.3f ms/pass' % (d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in d['kernel_time_ms'].items()})"
done
done 2>&1 | tee $T/ab_w6.txt
cp /tmp/librtgpu_base.so raytracer_amd/lib/librtgpu.so
touch raytracer_amd/lib/librtgpu.so raytracer_amd/lib/libraytracer_amd_host.so raytracer_amd/lib/rt_demo
python tools/vcm_bimodal.py --viewports 6 --regions 4 2>/dev/null | tee $T/vcm_bimodal.txt | cut -c1-400
