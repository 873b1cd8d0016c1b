cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s5
for i in 1 2; do
(bash tools/ab_env.sh "--workload bdpt-glass --steps 96 --warmup 8" RTGPU_VCM_CLASS=0 RTGPU_VCM_CLASS=3) >> gpurun_out/s5/ab_class96.txt 2>&1
(bash tools/ab_libs.sh "--workload bdpt-glass --steps 96 --warmup 8" base vl4 vl4c3 vl4c3n4 vc3) >> gpurun_out/s5/ab_waves96.txt 2>&1
done
cat gpurun_out/s5/ab_class96.txt gpurun_out/s5/ab_waves96.txt | cut -c1-200
