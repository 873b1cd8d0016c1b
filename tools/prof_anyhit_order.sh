#!/bin/bash
# Round 6: any-hit rays walk the FARTHEST entered child first (rt_trace_wide.inl) against nearest first (RTGPU_ANYHIT_FAR_FIRST=0), on one box.
#   bash tools/prof_anyhit_order.sh [out] [steps] [warmup]
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r06/anyhit_order_ab.txt}; STEPS=${2:-20}; WARM=${3:-5}
mkdir -p $(dirname $OUT)
echo "# python bench.py --steps $STEPS --warmup $WARM --no-pmc --no-cpu-baseline; two runs per setting, interleaved; $(date -u)" >> $OUT
for rep in 1 2; do for FAR in 0 1; do
  RTGPU_ANYHIT_FAR_FIRST=$FAR python bench.py --steps $STEPS --warmup $WARM --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('any-hit far first = $FAR  %8.1f Msamples/s  %.3f ms/pass (frame in HBM %.3f)  k_trace_wide class %.3f ms/launch  serial kernel ms %s' % (d['value'], d['ms_per_step'], d['frame_in_hbm']['ms_per_step'], d['roofline']['avg_launch_ms'], {k: round(v, 1) for k, v in d['kernel_time_ms'].items() if v}))
" >> $OUT
done; done
echo "# the walk's own counts (RTGPU_WIDE_DIAG=1, tools/wide_diag.py: 4 passes of the benchmark frame)" >> $OUT
for FAR in 0 1; do echo "any-hit far first = $FAR" >> $OUT; RTGPU_ANYHIT_FAR_FIRST=$FAR RTGPU_WIDE_DIAG=1 RTGPU_LANES=1 python tools/wide_diag.py 2>/dev/null | tail -6 | cut -c1-400 >> $OUT; done
cat $OUT
