cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-pmc --steps 48 --warmup 8"
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), d["kernel_time_ms"])'
for cfg in "28 32" "8 8" "16 16" "4 16" "16 4" "32 48" "8 24"; do
  set -- $cfg
  echo -n "refill=$1 other=$2 LDS: "; RTGPU_REFILL_MIN_IDLE=$1 RTGPU_OTHER_MIN_LANES=$2 $B 2>/dev/null | tail -1 | python -c "$P"
  echo -n "refill=$1 other=$2 noLDS: "; RTGPU_WIDE_NO_LDS=1 RTGPU_REFILL_MIN_IDLE=$1 RTGPU_OTHER_MIN_LANES=$2 $B 2>/dev/null | tail -1 | python -c "$P"
done
for bpc in 4 5 8; do echo -n "blocks/CU=$bpc noLDS r8 o8: "; RTGPU_WIDE_BLOCKS_PER_CU=$bpc RTGPU_WIDE_NO_LDS=1 RTGPU_REFILL_MIN_IDLE=8 RTGPU_OTHER_MIN_LANES=8 $B 2>/dev/null | tail -1 | python -c "$P"; done
mkdir -p gpurun_out/r02_wide0
i=0
for group in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  RTGPU_LANES=1 RTGPU_WIDE_NO_LDS=1 timeout 300 rocprofv3 --kernel-trace --pmc $group -d gpurun_out/r02_wide0/p$i -o r -- python bench.py --no-cpu-baseline --no-pmc --steps 24 --warmup 8 > /dev/null 2> gpurun_out/r02_wide0/err_p$i.txt
  db=$(find gpurun_out/r02_wide0/p$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py $db > gpurun_out/r02_wide0/pmc_$i.txt; fi
  rm -rf gpurun_out/r02_wide0/p$i
done
grep -h "k_trace_wide\|k_trace<24, false>" gpurun_out/r02_wide0/pmc_*.txt
python - <<'PY'
import sys, ctypes as C, numpy as np
sys.path.insert(0, "tests")
import kat_io, raytracer_amd as ra
lib = ra.rtgpu_lib(); c = C.c_void_p(); lib.rtgpu_create(0, C.byref(c))
func, inputs, expected = kat_io.load_kat("bsdf_sample.kat")
out = np.zeros_like(expected)
lib.rtgpu_kat(c, C.c_uint32(func), inputs.ctypes.data_as(C.c_void_p), C.c_uint32(inputs.shape[1]), out.ctypes.data_as(C.c_void_p), C.c_uint32(expected.shape[1]), C.c_uint32(len(inputs)))
bad = kat_io.bit_mismatch(expected, out)
rows = np.nonzero(bad.any(axis=1))[0]
print("bad rows", len(rows), "bsdf kinds", np.unique(inputs[rows, 12].view(np.uint32), return_counts=True))
for r in rows[:6]:
    print(r, "bsdf", inputs[r, 12].view(np.uint32), "in", inputs[r, :12], inputs[r, 16:23]); print("  exp", expected[r]); print("  got", out[r]); print("  cols", np.nonzero(bad[r])[0])
PY
