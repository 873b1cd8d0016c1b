cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in base dbl; do
  cp ab/librtgpu_$v.so raytracer_amd/lib/librtgpu.so
  echo -n "$v "; python bench.py --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['kernel_time_ms'], d['roofline']['avg_launch_ms'])"
done; done
cp ab/librtgpu_base.so raytracer_amd/lib/librtgpu.so
bash tools/pmc_diag.sh r02_diag0
