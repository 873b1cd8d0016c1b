cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/s5; mkdir -p $T
for i in 1 2 3 4 5 6 7 8; do
python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); kt = d.get('kernel_time_ms', {})
print('sponza20 %8.1f Msamples/s %7.3f ms/pass' % (d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in kt.items()})"
done > $T/variance_sponza20.txt
for i in 1 2 3 4 5 6; do
python bench.py --steps 256 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); kt = d.get('kernel_time_ms', {})
print('sponza256 %8.1f Msamples/s %7.3f ms/pass' % (d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in kt.items()})"
done > $T/variance_sponza256.txt
cat $T/variance_sponza20.txt $T/variance_sponza256.txt
rocm-smi --showclocks --showpower 2>/dev/null | head -30
