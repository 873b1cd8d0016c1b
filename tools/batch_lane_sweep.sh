cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/probes
for rep in 1 2; do for b in 4 5 6 7 10; do for l in 3 4 5; do
  RTGPU_PASS_BATCH_BASE=$b RTGPU_LANES=$l python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('base $b lanes $l %8.1f Msamples/s %7.3f ms/pass' % (d['value'], d['ms_per_step']))"
done; done; done | sort > gpurun_out/probes/batch_lane_sweep.txt; cat gpurun_out/probes/batch_lane_sweep.txt
