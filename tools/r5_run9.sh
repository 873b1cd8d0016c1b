# round 5, GPU call 9: config 5 with all arenas carved from one allocation (spread across contexts?) against an allocation per array
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05i
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_vcm.py -q -x 2>&1 | tail -3 | tee $T/pytest_vcm.log
echo "== one slab" | tee $T/vcm_slab.txt
python tools/vcm_bimodal.py --viewports 8 --regions 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['viewport'], d['Msamples_per_s_per_region'], d['serial_kernel_ms'])" | tee -a $T/vcm_slab.txt
echo "== one allocation per array (RTGPU_VCM_SLAB=0)" | tee -a $T/vcm_slab.txt
RTGPU_VCM_SLAB=0 python tools/vcm_bimodal.py --viewports 6 --regions 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['viewport'], d['Msamples_per_s_per_region'], d['serial_kernel_ms'])" | tee -a $T/vcm_slab.txt
