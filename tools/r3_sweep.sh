cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { env $1 python bench.py $2 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-44s %-28s %8.1f Msamples/s %7.3f ms/pass' % ('$1', '$2', d['value'], d['ms_per_step']))
"; }
for rep in 1 2; do
for e in "A=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8 RTGPU_LANES=5" "GPU_MAX_HW_QUEUES=8 RTGPU_LANES=6"; do
  run "$e" "--steps 20 --warmup 5"
  run "$e" "--steps 64 --warmup 5"
done
done
