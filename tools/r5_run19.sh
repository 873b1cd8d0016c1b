# round 5, GPU call 19: with the empty monster launches gone -- the re-trace folded into the launch (block-local) and one traversal block per CU less, on the full frame
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05s
mkdir -p $T
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_DUMMY=0 RTGPU_LOCAL_EXACT=1 RTGPU_TRAV_BLOCKS_PER_CU=4 "RTGPU_TRAV_BLOCKS_PER_CU=4 RTGPU_LOCAL_EXACT=1" RTGPU_RETRACE_MONSTERS=1 2>&1 | tee $T/ab.txt
bash tools/ab_env.sh "--steps 64 --warmup 5" RTGPU_DUMMY=0 RTGPU_LOCAL_EXACT=1 RTGPU_TRAV_BLOCKS_PER_CU=4 2>&1 | tee -a $T/ab.txt
bash tools/prof_concurrency.sh > $T/concurrency.txt 2>&1; head -12 $T/concurrency.txt
