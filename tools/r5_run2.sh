# round 5, GPU call 2: leaf phase with parked quotients + gate fetched with the triangles (16-entry stack), A/B against the previous library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=gpurun_out/r05b
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kat.py -q -x 2>&1 | tail -5 > $T/pytest_parity.log
tail -3 $T/pytest_parity.log
bash tools/ab_libs.sh "--steps 20 --warmup 5" r05a base 2>&1 | tee $T/ab_leaf.txt
bash tools/ab_libs.sh "--steps 64 --warmup 5" r05a base 2>&1 | tee -a $T/ab_leaf.txt
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_OTHER_MIN_LANES=24 RTGPU_OTHER_MIN_LANES=32 RTGPU_OTHER_MIN_LANES=40 RTGPU_REFILL_MIN_IDLE=20 RTGPU_REFILL_MIN_IDLE=36 2>&1 | tee $T/ab_knobs.txt
for b in 3 4 5 6; do RTGPU_LANES=1 RTGPU_TRAV_BLOCKS_PER_CU=$b python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('serial lanes, trav blocks per CU $b: %.1f Msamples/s' % d['value'], d['kernel_time_ms'])"; done 2>&1 | tee $T/serial_blocks_per_cu.txt
RTGPU_WIDE_DIAG=1 python tools/wide_diag.py > $T/wide_diag_1.txt 2>&1; tail -6 $T/wide_diag_1.txt
