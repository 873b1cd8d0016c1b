# round 5, GPU call 15: small-frame policy re-swept on the round's kernels (block-local re-trace on / off, tail hand-over bounce) for 1/8, 1/4, 1/2 shards at 20 passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05o
mkdir -p $T
for n in 8 4 2; do
  echo "== shard 1/$n"
  BENCH_EMULATE_SHARD=$n bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_DUMMY=0 RTGPU_LOCAL_EXACT=0 RTGPU_LOCAL_EXACT=1 "RTGPU_LOCAL_EXACT=0 RTGPU_TAIL_DEPTH=0" "RTGPU_LOCAL_EXACT=1 RTGPU_TAIL_DEPTH=0" RTGPU_TAIL_DEPTH=4 RTGPU_TAIL_DEPTH=6 2>&1 | cut -c1-200
done | tee $T/shard_policy.txt
