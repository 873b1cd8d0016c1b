# round 5, GPU call 6: the degenerate-axis prune of the binary walk (light yaw 0 = sponza.json's orientation against the 20 degrees of rounds 1-4) + FETCH_SIZE calibration
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05f
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -q -x 2>&1 | tail -5 > $T/pytest_parity.log
tail -3 $T/pytest_parity.log
for yaw in 0 20; do
  for lib in r05a base; do
    if [ $lib = base ]; then cp /tmp/keep.so raytracer_amd/lib/librtgpu.so 2>/dev/null; else cp raytracer_amd/lib/librtgpu.so /tmp/keep.so; cp variants/librtgpu_$lib.so raytracer_amd/lib/librtgpu.so; fi
    touch raytracer_amd/lib/librtgpu.so raytracer_amd/lib/libraytracer_amd_host.so raytracer_amd/lib/rt_demo
    for rep in 1 2; do
    BENCH_LIGHT_YAW=$yaw python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('yaw $yaw  lib $lib: %.1f Msamples/s, %.3f ms/pass' % (d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in d['kernel_time_ms'].items()}, 'image mean', d['image']['mean_per_pass'])"
    done
  done
done 2>&1 | tee $T/yaw_ab.txt
cp /tmp/keep.so raytracer_amd/lib/librtgpu.so
bash tools/fetch_calib.sh 2>&1 | tail -12
