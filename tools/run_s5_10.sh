cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; T=gpurun_out/s5; mkdir -p $T
RTGPU_LANES=1 rocprofv3 --kernel-trace --stats -d $T/prof_pk -o r -- python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline > /dev/null 2>&1
db=$(find $T/prof_pk -name '*.db' | head -1); python tools/rocpd_summary.py $db | head -8 | cut -c1-150; python tools/rocpd_summary.py --timeline $db | grep -A3 "k_generate_dense" | head -24 | cut -c1-90; rm -rf $T/prof_pk
for b in 4 6 12; do RTGPU_PACKET_BLOCKS=$b python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); kt = d.get('kernel_time_ms', {})
print('blocks $b %8.1f Msamples/s trace %.2f' % (d['value'], kt.get('trace', 0)))"; done
