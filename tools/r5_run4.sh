# round 5, GPU call 4: leaf records (one 128-byte line per leaf visit) against the separate triangle / box arrays, with and without the register diet
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05d
mkdir -p $T
cp raytracer_amd/lib/librtgpu.so /tmp/keep.so
cp variants/librtgpu_vg.so raytracer_amd/lib/librtgpu.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wide or packet or full_size or retrace or tail" 2>&1 | tail -30 > $T/pytest_vg.log
tail -4 $T/pytest_vg.log
cp /tmp/keep.so raytracer_amd/lib/librtgpu.so
if grep -q "passed" $T/pytest_vg.log && ! grep -q "failed\|error" $T/pytest_vg.log; then
  bash tools/ab_libs.sh "--steps 20 --warmup 5" r05a vc vg vh vi 2>&1 | tee $T/ab_variants.txt
  bash tools/ab_libs.sh "--steps 64 --warmup 5" r05a vc vg 2>&1 | tee -a $T/ab_variants.txt
else
  bash tools/ab_libs.sh "--steps 20 --warmup 5" r05a vc 2>&1 | tee $T/ab_variants.txt
fi
