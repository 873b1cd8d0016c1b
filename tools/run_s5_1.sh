cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s5
(timeout 900 python -m pytest tests/test_gpu_vcm.py -x -q 2>&1 | tail -5) > gpurun_out/s5/vcm_tests.log
(bash tools/ab_env.sh "--workload bdpt-glass --steps 20 --warmup 5" RTGPU_VCM_CLASS=0 RTGPU_VCM_CLASS=3) > gpurun_out/s5/ab_class.txt 2>&1
(bash tools/ab_libs.sh "--workload bdpt-glass --steps 20 --warmup 5" base vl4 vl4c3 vl4c3n4 vc3) > gpurun_out/s5/ab_waves.txt 2>&1
cat gpurun_out/s5/vcm_tests.log gpurun_out/s5/ab_class.txt gpurun_out/s5/ab_waves.txt
