# round 5, GPU call 26: k_shade_dense ranks a chunk's survivors by ray bin (RTGPU_DENSE_SORT: 0 arrival order, 1 direction octant, 2 octant x major axis, 3 octant x origin half spaces):
# the traversal kernels take their rays in record order, so what lies side by side walks side by side.  Parity under mode 3, then the A/B on the driver's command.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05x
mkdir -p $T
RTGPU_DENSE_SORT=3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $T/parity_sort3.txt
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_DENSE_SORT=0 RTGPU_DENSE_SORT=1 RTGPU_DENSE_SORT=2 RTGPU_DENSE_SORT=3 2>&1 | tee $T/ab_dense_sort.txt
bash tools/ab_env.sh "--steps 128 --warmup 5" RTGPU_DENSE_SORT=0 RTGPU_DENSE_SORT=3 2>&1 | tee -a $T/ab_dense_sort.txt
