cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quant or dense" 2>&1 | tail -4
cp raytracer_amd/lib/librtgpu.so /tmp/keep.so
B="python bench.py --no-cpu-baseline --no-pmc --steps 64 --warmup 8"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), d["kernel_time_ms"])'
for rep in 1 2; do for v in base oct; do
  cp ab/librtgpu_$v.so raytracer_amd/lib/librtgpu.so
  echo -n "$v: "; $B 2>/dev/null | tail -1 | python -c "$P"
done; done
cp /tmp/keep.so raytracer_amd/lib/librtgpu.so
for cfg in "16 16" "28 24" "24 32" "32 32" "28 40"; do set -- $cfg; echo -n "refill=$1 other=$2: "; RTGPU_REFILL_MIN_IDLE=$1 RTGPU_OTHER_MIN_LANES=$2 $B 2>/dev/null | tail -1 | python -c "$P"; done
for l in 2 4; do echo -n "lanes=$l: "; RTGPU_LANES=$l $B 2>/dev/null | tail -1 | python -c "$P"; done
for sb in 4 16; do echo -n "shade blocks/CU=$sb: "; RTGPU_SHADE_BLOCKS_PER_CU=$sb $B 2>/dev/null | tail -1 | python -c "$P"; done
echo "--- driver-like default run with in-run PMC and CPU baseline"
/usr/bin/time -v python bench.py --steps 20 --warmup 5 2> gpurun_out/bench_r02_err.txt | tail -1 > gpurun_out/bench_r02_driverlike.json
grep "Elapsed (wall" gpurun_out/bench_r02_err.txt
python -c "
import json; d=json.load(open('gpurun_out/bench_r02_driverlike.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'], indent=1)); print(d.get('traffic_per_launch')); print(d['cpu_baseline'])"
