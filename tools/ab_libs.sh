# A/B of library variants on one box: bash tools/ab_libs.sh "<bench args>" name1 name2 ...   (names of variants/librtgpu_<name>.so; "base" = the tree's own build)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ARGS="$1"; shift
cp raytracer_amd/lib/librtgpu.so /tmp/librtgpu_base.so
for rep in 1 2; do
for name in "$@"; do
  if [ $name = base ]; then cp /tmp/librtgpu_base.so raytracer_amd/lib/librtgpu.so; else cp variants/librtgpu_$name.so raytracer_amd/lib/librtgpu.so; fi
  touch raytracer_amd/lib/librtgpu.so raytracer_amd/lib/libraytracer_amd_host.so raytracer_amd/lib/rt_demo
  python bench.py $ARGS --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
kt = d.get('kernel_time_ms', {})
print('%-14s %s: %.1f Msamples/s, %.3f ms/pass' % ('$name', '$ARGS', d['value'], d['ms_per_step']), {k: round(v, 2) for k, v in kt.items()} if isinstance(kt, dict) else '')
"
done
done
cp /tmp/librtgpu_base.so raytracer_amd/lib/librtgpu.so
