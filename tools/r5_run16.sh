# round 5, GPU call 16: k_trace_wide2's diet (per-phase selectors, wave-level tallies: scratch 104 -> 44 B) on two-level scenes; the small-frame policy after the sweep
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05p
mkdir -p $T
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_images.py tests/test_gpu_multi_device.py -q -x 2>&1 | tail -3 | tee $T/pytest.log
bash tools/ab_libs.sh "--workload cornell --steps 32 --warmup 4" r05p base 2>&1 | tee $T/ab_wide2.txt
bash tools/ab_libs.sh "--workload zoo --steps 32 --warmup 4" r05p base 2>&1 | tee -a $T/ab_wide2.txt
bash tools/ab_libs.sh "--workload cornell --width 640 --height 480 --depth 4 --steps 16 --warmup 4" r05p base 2>&1 | tee -a $T/ab_wide2.txt
for n in 8 4; do BENCH_EMULATE_SHARD=$n bash tools/ab_libs.sh "--steps 20 --warmup 5" r05p base 2>&1 | sed "s/^/shard 1\/$n  /"; done | tee $T/ab_shard_policy.txt
