# A/B of environment settings on one box (two repetitions, interleaved):  bash tools/ab_env.sh "<bench.py args>" "<env A>" "<env B>" ...
# e.g.  gpurun -- 'bash tools/ab_env.sh "--workload cornell --steps 32 --warmup 4" RTGPU_WIDE2=0 RTGPU_WIDE2=1'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ARGS="$1"; shift
for rep in 1 2; do
for E in "$@"; do
  env $E python bench.py $ARGS --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
kt = d.get('kernel_time_ms', {})
print('%-36s %-60s %8.1f Msamples/s %7.3f ms/pass' % ('$E', '$ARGS', d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in kt.items()} if isinstance(kt, dict) else '')
"
done
done
