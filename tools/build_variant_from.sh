#!/bin/bash
# builds variants/librtgpu_<name>.so from another source tree (e.g. a git worktree of an earlier commit):  tools/build_variant_from.sh name /path/to/tree
set -e
name=$1; src=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/variant_$name; rm -rf $W; mkdir -p $W
mkdir -p $ROOT/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden"
OBJS=""
for u in rt_shade rt_tail; do
  if [ -f $src/raytracer_amd/csrc/$u.hip ]; then /opt/rocm/bin/hipcc $F -mllvm -simplifycfg-sink-common=false -c $src/raytracer_amd/csrc/$u.hip -o $W/$u.o & OBJS="$OBJS $W/$u.o"; fi
done
/opt/rocm/bin/hipcc $F -c $src/raytracer_amd/csrc/rt_trace.hip -o $W/rt_trace.o &
/opt/rocm/bin/hipcc $F -c $src/raytracer_amd/csrc/rt_runtime.hip -o $W/rt_runtime.o
wait
/opt/rocm/bin/hipcc $F -shared $W/rt_runtime.o $W/rt_trace.o $OBJS $src/raytracer_amd/csrc/rt_vcm_photons.hip -o $ROOT/variants/librtgpu_$name.so
echo built variants/librtgpu_$name.so
