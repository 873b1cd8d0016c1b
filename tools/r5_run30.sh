# round 5, GPU call 30: batch lanes on streams of different priorities (RTGPU_LANE_PRIORITY: 0 none, 1 high / normal / normal / low in creation order, 2 all high) -- DESIGN 8's "not tried: stream priorities"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05x
mkdir -p $T
RTGPU_LANE_PRIORITY=1 RTGPU_VERBOSE=1 python bench.py --steps 4 --warmup 2 --no-pmc --no-cpu-baseline 2>&1 | grep "stream .: priority" | head -8 | tee $T/ab_lane_priority.txt
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_LANE_PRIORITY=0 RTGPU_LANE_PRIORITY=1 RTGPU_LANE_PRIORITY=2 2>&1 | tee -a $T/ab_lane_priority.txt
bash tools/ab_env.sh "--steps 128 --warmup 5" RTGPU_LANE_PRIORITY=0 RTGPU_LANE_PRIORITY=1 2>&1 | tee -a $T/ab_lane_priority.txt
