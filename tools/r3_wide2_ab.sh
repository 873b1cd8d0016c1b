cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { env $1 python bench.py $2 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
kt = d.get('kernel_time_ms', {})
print('%-14s %-74s %8.1f Msamples/s %7.3f ms/pass' % ('$1', '$2', d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in kt.items()} if isinstance(kt, dict) else '')
"; }
for rep in 1 2; do
for e in RTGPU_WIDE2=0 RTGPU_WIDE2=1; do
  run $e "--workload cornell --width 640 --height 480 --depth 4 --steps 16 --warmup 4"
  run $e "--workload cornell --steps 32 --warmup 4"
  run $e "--workload zoo --steps 32 --warmup 4"
  run $e "--workload sphere --steps 32 --warmup 4"
done
done
