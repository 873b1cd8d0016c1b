# round 5, GPU call 12: refill tweaks of k_trace_wide (no request division under Single; the inverse transform fetched per refill by scalar loads: the loop header no longer spills scalar registers)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05l
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "wide or packet or full_size or retrace or tail or axis" 2>&1 | tail -3 | tee $T/pytest.log
bash tools/ab_libs.sh "--steps 20 --warmup 5" r05m base r05m base 2>&1 | tee $T/ab_refill.txt
bash tools/ab_libs.sh "--steps 64 --warmup 5" r05m base 2>&1 | tee -a $T/ab_refill.txt
