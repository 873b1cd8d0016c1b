cd $GRAFT_REPO_ROOT
cp raytracer_amd/lib/librtgpu.so /tmp/librtgpu_base.so
try() { # name env
  for k in 1 2 3; do
    env $2 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:faulthandler > /tmp/out.txt 2>&1; rc=$?
    echo "$1 [$2] run $k: rc=$rc $(tail -c 120 /tmp/out.txt | tr '\n' ' ')"
  done
}
try current A=1
try current-nosort RTGPU_SHADE_SORT=0
cp variants/librtgpu_head.so raytracer_amd/lib/librtgpu.so; touch raytracer_amd/lib/*
try head A=1
cp variants/librtgpu_stage.so raytracer_amd/lib/librtgpu.so; touch raytracer_amd/lib/*
try stage A=1
cp /tmp/librtgpu_base.so raytracer_amd/lib/librtgpu.so
