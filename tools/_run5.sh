cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quant or lds_staged or single_object or mesh_two_level or full_size_sponza" 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --no-pmc --steps 64 --warmup 8"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), d["kernel_time_ms"])'
for rep in 1 2; do
echo -n "default (LDS top): "; $B 2>/dev/null | tail -1 | python -c "$P"
echo -n "NO_LDS_TOP: "; RTGPU_NO_LDS_TOP=1 $B 2>/dev/null | tail -1 | python -c "$P"
done
echo -n "QUANT: "; RTGPU_QUANT=1 $B 2>/dev/null | tail -1 | python -c "$P"
