#!/bin/bash
# Round 6: the streaming policy (first batch size x lanes) re-swept on this round's kernels at the driver's 20 passes.
#   bash tools/batch_policy_sweep.sh [out] [steps] [warmup]
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=${1:-gpurun_out/r06/batch_policy_sweep.txt}; STEPS=${2:-20}; WARM=${3:-5}
mkdir -p $(dirname $OUT)
echo "# python bench.py --steps $STEPS --warmup $WARM --no-pmc --no-cpu-baseline; $(date -u)" >> $OUT
for LANES in 3 4 5 6; do for BASE in 4 5 7 10; do for rep in 1 2; do
  RTGPU_LANES=$LANES RTGPU_PASS_BATCH_BASE=$BASE python bench.py --steps $STEPS --warmup $WARM --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('lanes %d  first batch %2d  %8.1f Msamples/s  %.3f ms/pass' % ($LANES, $BASE, d['value'], d['ms_per_step']))
" >> $OUT
done; done; done
cat $OUT
