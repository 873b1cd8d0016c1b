# round 5, GPU call 20: batch lanes re-swept on the final library (4 is the default)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05u
mkdir -p $T
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_LANES=4 RTGPU_LANES=5 RTGPU_LANES=6 RTGPU_LANES=3 2>&1 | tee $T/ab_lanes.txt
bash tools/ab_env.sh "--steps 64 --warmup 5" RTGPU_LANES=4 RTGPU_LANES=5 RTGPU_LANES=6 2>&1 | tee -a $T/ab_lanes.txt
