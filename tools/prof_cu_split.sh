#!/bin/bash
# Round 6, review item 3: CU-partitioned streams (RTGPU_CU_SPLIT) against the default lane overlap, on one box, at the driver's command.  The experiment's code is NOT in the
# library any more (measured slower everywhere): apply profiles/r06_cu_split_experiment.patch to raytracer_amd/csrc/rt_runtime.hip and rebuild first.
#   bash tools/prof_cu_split.sh [out=gpurun_out/r06/cu_split_ab.txt] [steps=20] [warmup=5]
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r06/cu_split_ab.txt}; STEPS=${2:-20}; WARM=${3:-5}
mkdir -p $(dirname $OUT)
run() {   # label, env assignments...
  label=$1; shift
  for rep in 1 2; do
    env "$@" python bench.py --steps $STEPS --warmup $WARM --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-46s %8.1f Msamples/s  %.3f ms/pass (frame in HBM %.3f)  serial kernel ms %s' % ('$label', d['value'], d['ms_per_step'], d['frame_in_hbm']['ms_per_step'], {k: round(v, 1) for k, v in d['kernel_time_ms'].items() if v}))
" >> $OUT
  done
}
echo "# python bench.py --steps $STEPS --warmup $WARM --no-pmc --no-cpu-baseline; two runs per setting; $(date -u)" >> $OUT
run "default (no partition)" RTGPU_VERBOSE=0
for K in 16 32 64; do
  for LAYOUT in 0 1; do
    run "split K=$K layout=$LAYOUT short kernels confined" RTGPU_CU_SPLIT=$K RTGPU_CU_SPLIT_LAYOUT=$LAYOUT
    run "split K=$K layout=$LAYOUT short kernels unconfined" RTGPU_CU_SPLIT=$K RTGPU_CU_SPLIT_LAYOUT=$LAYOUT RTGPU_CU_SPLIT_AUX=0
  done
done
run "default (no partition), again" RTGPU_VERBOSE=0
cat $OUT
