cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -6
SECONDS=0
python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_r02_err.txt | tail -1 > gpurun_out/bench_r02_driverlike.json
echo "wall $SECONDS s"; tail -3 gpurun_out/bench_r02_err.txt
python -c "
import json; d=json.load(open('gpurun_out/bench_r02_driverlike.json')); print(d['value'], d['ms_per_step']); r=d['roofline']; print({k: r[k] for k in ('kernel','launches','avg_launch_ms','achieved','frac','traffic','algorithmic_bytes_per_launch','traffic_over_algorithmic','algorithmic_frac_of_l2_peak')}); print(d['cpu_baseline'])"
