# round 5, GPU call 18: the empty k_trace_monster launches behind every re-trace launch wait for a CU slot under concurrency (27.7 ms summed in the timed region, profiles/r05_concurrency.txt): off / on
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05r
mkdir -p $T
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_RETRACE_MONSTERS=1 RTGPU_RETRACE_MONSTERS=0 RTGPU_RETRACE_MONSTERS=1 RTGPU_RETRACE_MONSTERS=0 2>&1 | tee $T/ab_monsters.txt
bash tools/ab_env.sh "--steps 64 --warmup 5" RTGPU_RETRACE_MONSTERS=1 RTGPU_RETRACE_MONSTERS=0 2>&1 | tee -a $T/ab_monsters.txt
