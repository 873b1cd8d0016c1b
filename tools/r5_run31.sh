# round 5, GPU call 31: a 1/8 shard at the driver's 20 passes against the traversal grid (blocks per CU) and the smallest claim (its launches are half drain: fewer, longer-lived waves or finer claims?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05x
mkdir -p $T
BENCH_EMULATE_SHARD=8 bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_DUMMY=0 RTGPU_TRAV_BLOCKS_PER_CU=3 RTGPU_TRAV_BLOCKS_PER_CU=4 RTGPU_WIDE_CHUNK_MIN=16 RTGPU_WIDE_CHUNK_MIN=32 RTGPU_WIDE_CHUNK_MIN=128 "RTGPU_TRAV_BLOCKS_PER_CU=4 RTGPU_WIDE_CHUNK_MIN=32" RTGPU_TAIL_BLOCKS_PER_CU=3 RTGPU_TAIL_BLOCKS_PER_CU=5 RTGPU_SHADE_BLOCKS_PER_CU=4 RTGPU_SHADE_BLOCKS_PER_CU=16 2>&1 | tee $T/ab_shard_grid.txt
