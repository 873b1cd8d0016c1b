cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "--- driver-like default run with in-run PMC and CPU baseline"
SECONDS=0
python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_r02_err.txt | tail -1 > gpurun_out/bench_r02_driverlike.json
echo "wall $SECONDS s"; tail -5 gpurun_out/bench_r02_err.txt
python -c "
import json; d=json.load(open('gpurun_out/bench_r02_driverlike.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'], indent=1)); print(d.get('traffic_per_launch')); print(d['cpu_baseline'])"
SECONDS=0
python bench.py --workload bdpt-glass --steps 16 --warmup 4 2> gpurun_out/bench_r02_bdpt_err.txt | tail -1 > gpurun_out/bench_r02_bdpt.json
echo "wall $SECONDS s"; tail -3 gpurun_out/bench_r02_bdpt_err.txt; python -c "
import json; d=json.load(open('gpurun_out/bench_r02_bdpt.json')); print(d['value'], d['ms_per_step'], d['image']); print(json.dumps(d['roofline'], indent=1)); print(d['cpu_baseline'])"
