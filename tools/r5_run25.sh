# round 5, GPU call 25: the whole GPU suite on the library with the reversed claim order; the two-level walk (k_trace_wide2) in both orders; the driver's command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05w
mkdir -p $T
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $T/gpu_suite.txt
bash tools/ab_env.sh "--workload cornell --steps 32 --warmup 4" RTGPU_WIDE_REVERSE=0 RTGPU_WIDE_REVERSE=1 2>&1 | tee $T/ab_reverse_wide2.txt
bash tools/ab_env.sh "--workload zoo --steps 32 --warmup 4" RTGPU_WIDE_REVERSE=0 RTGPU_WIDE_REVERSE=1 2>&1 | tee -a $T/ab_reverse_wide2.txt
python bench.py --steps 20 --warmup 5 > $T/bench_steps20.json 2> $T/bench_steps20.err
tail -1 $T/bench_steps20.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print(round(d['value'], 1), d['steps'], round(d['ms_per_step'], 3), r['avg_launch_ms'], r['frac'], r['traffic_frac'], r.get('compulsory_bytes_per_launch'), r.get('traffic_over_compulsory'), r['bound'], d['cpu_baseline']['value'])"
head -c 200 $T/bench_steps20.json; echo
