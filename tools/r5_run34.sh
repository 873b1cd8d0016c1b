# round 5, GPU call 34: WHAT in the record order the next shading launch is sensitive to -- the ray-bin patch (profiles/r05_ray_bin_order_experiment.patch) as a variant library with
# control orders: RTGPU_DENSE_SORT = 0 arrival order, 6 the chunk's waves in their own order, 4 in reverse order (both: every wave's paths stay together), 5 every output wave takes every
# fourth path (spans all four 8 x 8 tiles of the chunk, no direction sorting), 7 rows of the tiles regrouped, 1 direction octant (call 26)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
T=gpurun_out/r05x
mkdir -p $T
cp raytracer_amd/lib/librtgpu.so /tmp/librtgpu_base.so
cp variants/librtgpu_binorder.so raytracer_amd/lib/librtgpu.so
touch raytracer_amd/lib/librtgpu.so raytracer_amd/lib/libraytracer_amd_host.so raytracer_amd/lib/rt_demo
bash tools/ab_env.sh "--steps 20 --warmup 5" RTGPU_DENSE_SORT=0 RTGPU_DENSE_SORT=6 RTGPU_DENSE_SORT=4 RTGPU_DENSE_SORT=5 RTGPU_DENSE_SORT=7 RTGPU_DENSE_SORT=1 2>&1 | tee $T/ab_order_controls.txt
cp /tmp/librtgpu_base.so raytracer_amd/lib/librtgpu.so
