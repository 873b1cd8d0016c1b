# round 5, GPU call 24: 1/8 shard at the driver's 20 passes -- passes per batch re-swept (20 = one batch, nothing to overlap its drains with; 10 / 7 / 5 = 2 / 3 / 4 batches on the lanes), claim order reversed
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05w
mkdir -p $T
export RTGPU_WIDE_REVERSE=1
BENCH_EMULATE_SHARD=8 bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_SMALL_FRAME_BATCH=20 RTGPU_SMALL_FRAME_BATCH=10 RTGPU_SMALL_FRAME_BATCH=7 RTGPU_SMALL_FRAME_BATCH=5 "RTGPU_SMALL_FRAME_BATCH=10 RTGPU_LANES=2" 2>&1 | tee $T/ab_shard_batch.txt
BENCH_EMULATE_SHARD=4 bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_WIDE_REVERSE=0 RTGPU_WIDE_REVERSE=1 2>&1 | tee -a $T/ab_shard_batch.txt
