# round 5, GPU call 21: `python bench.py` without flags (256 passes, 16 warm-up: BASELINE config 3's spp) -- completes, and how long it takes end to end
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05v
mkdir -p $T
start=$(date +%s)
python bench.py > $T/bench_noflags.json 2> $T/bench_noflags.err
end=$(date +%s)
echo "wall seconds: $((end-start))"
tail -1 $T/bench_noflags.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print(round(d['value'], 1), d['steps'], d['warmup'], round(d['ms_per_step'], 3), r['avg_launch_ms'], r['frac'], r['traffic_frac'], r['bound'], d['cpu_baseline']['value'])"
tail -2 $T/bench_noflags.err
