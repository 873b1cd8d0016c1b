# round 5, GPU call 10: do same-slot accesses of the record arrays collide?  record stride padded (PathTracerMIS), slab pieces skewed (config 5)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05j
mkdir -p $T
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_ARENA_PAD=0 RTGPU_ARENA_PAD=16 RTGPU_ARENA_PAD=272 RTGPU_ARENA_PAD=4112 RTGPU_ARENA_PAD=65552 2>&1 | tee $T/ab_arena_pad.txt
for skew in 0 4352 69888; do
echo "== slab skew $skew" | tee -a $T/vcm_skew.txt
RTGPU_VCM_SLAB_SKEW=$skew python tools/vcm_bimodal.py --viewports 5 --regions 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['viewport'], d['Msamples_per_s_per_region'], d['serial_kernel_ms'])" | tee -a $T/vcm_skew.txt
done
