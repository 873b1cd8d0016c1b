# same-box A/B of library variants: ab/librtgpu_<name>.so for every name given (ab/ is git-ignored and travels with gpurun)
cd $GRAFT_REPO_ROOT
cp raytracer_amd/lib/librtgpu.so /tmp/librtgpu_keep.so
for rep in 1 2; do
for v in "$@"; do
  cp ab/librtgpu_$v.so raytracer_amd/lib/librtgpu.so
  echo -n "$v "; RTGPU_TRAV_BLOCKS_PER_CU=${BLOCKS:-0} python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['kernel_time_ms'])"
done; done
cp /tmp/librtgpu_keep.so raytracer_amd/lib/librtgpu.so
