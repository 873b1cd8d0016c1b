cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { env $1 python bench.py $2 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
kt = d.get('kernel_time_ms', {})
print('%-22s %-50s %8.1f Msamples/s %7.3f ms/pass' % ('$1', '$2', d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in kt.items()} if isinstance(kt, dict) else '')
"; }
for rep in 1 2; do
  run RTGPU_NO_DENSE=1 "--workload sponza-all --steps 20 --warmup 5"
  run RTGPU_NO_DENSE=0 "--workload sponza-all --steps 20 --warmup 5"
  run RTGPU_NO_DENSE=1 "--workload sponza-all --steps 64 --warmup 5"
  run RTGPU_NO_DENSE=0 "--workload sponza-all --steps 64 --warmup 5"
done
