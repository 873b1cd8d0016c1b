# A/B on one box: bash tools/r3_ab.sh "<env A>" "<env B>" [steps...]   (an env may be empty)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r3ab
mkdir -p $T
A="$1"; B="$2"; shift; shift
STEPS="${@:-20 64}"
for rep in 1 2; do
for steps in $STEPS; do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --steps $steps --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
kt = d.get('kernel_time_ms', {})
print('$tag [$E] steps $steps: %.1f Msamples/s, %.3f ms/pass' % (d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in kt.items()} if isinstance(kt, dict) else '')
"
  done
done
done
