# round 5, GPU call 3: which of the leaf-phase changes costs / pays -- isolation variants (tools: /tmp/mkvar.sh; RT_WIDE_GATE_MODE / RT_WIDE_DIET / RT_CHUNK_SCALAR)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r05c
mkdir -p $T
cp raytracer_amd/lib/librtgpu.so /tmp/keep.so
cp variants/librtgpu_vd.so raytracer_amd/lib/librtgpu.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wide or packet or full_size or retrace or tail" 2>&1 | tail -3 | tee $T/pytest_vd.log
cp /tmp/keep.so raytracer_amd/lib/librtgpu.so
bash tools/ab_libs.sh "--steps 20 --warmup 5" r05a va vb vc vd ve vf base 2>&1 | tee $T/ab_variants.txt
