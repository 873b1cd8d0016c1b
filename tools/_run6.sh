cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8
B="python bench.py --no-cpu-baseline --no-pmc --steps 64 --warmup 8"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), d["kernel_time_ms"])'
for rep in 1 2; do
echo -n "dense: "; $B 2>/dev/null | tail -1 | python -c "$P"
echo -n "NO_DENSE: "; RTGPU_NO_DENSE=1 $B 2>/dev/null | tail -1 | python -c "$P"
done
