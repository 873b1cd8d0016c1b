cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quant or single_object or full_size_sponza" 2>&1 | tail -15
python -m pytest tests/test_gpu_kat.py -x -q -m gpu 2>&1 | tail -15
for v in 0 1; do
echo -n "NO_QUANT=$v "; RTGPU_NO_QUANT=$v python bench.py --no-cpu-baseline --no-pmc --steps 64 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['kernel_time_ms'], d.get('kernel_launches'), d['roofline']['avg_launch_ms'])"
done
