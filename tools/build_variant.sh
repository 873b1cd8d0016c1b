#!/bin/bash
# builds variants/librtgpu_<name>.so from a copy of the csrc tree after applying sed scripts:  tools/build_variant.sh name [file 'sed-expr']...
set -e
name=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/variant_$name; rm -rf $W; mkdir -p $W/raytracer_amd
cp -r $ROOT/raytracer_amd/csrc $W/raytracer_amd/csrc; cp -r $ROOT/include $W/include
while [ $# -gt 0 ]; do f=$1; e=$2; shift; shift; sed -i "$e" $W/raytracer_amd/csrc/$f; done
mkdir -p $ROOT/variants
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden"
OBJS=""
for u in rt_shade rt_tail; do
  if [ -f $W/raytracer_amd/csrc/$u.hip ]; then /opt/rocm/bin/hipcc $F -mllvm -simplifycfg-sink-common=false -c $W/raytracer_amd/csrc/$u.hip -o $W/$u.o & OBJS="$OBJS $W/$u.o"; fi
done
/opt/rocm/bin/hipcc $F -c $W/raytracer_amd/csrc/rt_trace.hip -o $W/rt_trace.o &
/opt/rocm/bin/hipcc $F -c $W/raytracer_amd/csrc/rt_runtime.hip -o $W/rt_runtime.o
wait
/opt/rocm/bin/hipcc $F -shared $W/rt_runtime.o $W/rt_trace.o $OBJS $W/raytracer_amd/csrc/rt_vcm_photons.hip -o $ROOT/variants/librtgpu_$name.so
echo built variants/librtgpu_$name.so
