# 2-D sweep of the traversal kernels' scheduling knobs, two runs each (gpurun -- 'bash tools/knob_sweep.sh [steps] [warmup] [out]'; profiles/r04_interior_step_probes.txt, section 4;
# round 6: re-swept behind the far-first any-hit order, profiles/r06_knob_sweep.txt)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
STEPS=${1:-64}; WARM=${2:-5}; OUT=${3:-gpurun_out/probes/knob_sweep.txt}; mkdir -p $(dirname $OUT)
ENVS=""
for r in 12 20 28 36; do for o in 16 24 32 40; do ENVS="$ENVS RTGPU_REFILL_MIN_IDLE=$r,RTGPU_OTHER_MIN_LANES=$o"; done; done
for rep in 1 2; do for E in $ENVS; do
  env $(echo $E | tr ',' ' ') python bench.py --steps $STEPS --warmup $WARM --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); kt = d.get('kernel_time_ms', {})
print('%-52s %8.1f Msamples/s %7.3f ms/pass trace %.1f' % ('$E', d['value'], d['ms_per_step'], kt.get('trace', 0)))"
done; done > $OUT
sort $OUT | cut -c1-110
