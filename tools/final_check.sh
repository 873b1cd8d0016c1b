# last GPU call of a round: the suite, smoke() and the driver's command on the FINAL library (gpurun -- 'bash tools/final_check.sh [round tag]'); summaries -> gpurun_out/<tag>final/
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/${1:-r06}final
mkdir -p $T
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$" | grep "passed\|failed\|error\|^\." > $T/pytest_gpu.log
tail -2 $T/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "^smoke" > $T/smoke.log; cat $T/smoke.log
python bench.py --steps 20 --warmup 5 > $T/bench_default_steps20.json 2> $T/bench_default_steps20.err
tail -1 $T/bench_default_steps20.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s', round(d['ms_per_step'], 3), 'ms/pass; frame in HBM', round(d['frame_in_hbm']['value'], 1), '; trace class', round(r['avg_launch_ms'], 4), 'ms/launch; frac', round(r['frac'], 4), 'request rate', round(r['request_rate_over_hbm_peak'], 3), 'binding', r['binding_ceiling']['name'], round(r['frac_of_binding_ceiling'], 3), d['kernel_time_ms'])"
