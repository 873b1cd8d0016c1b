// rt_trace.hip -- the traversal kernels of the library and the small kernels around them, a translation unit of their own:
//   rt_trace_binary.inl   k_trace (the reference's walk: binary BVH, near child first), k_trace_monster
//   rt_wide_grid.inl    the 16-bit grid, the leaf gates and the exactness argument the 4-wide walks share
//   rt_trace_wide.inl     k_trace_wide (4-wide tree of a single-mesh scene; the rays it does not decide go through the reference's walk in the same launch)
//   rt_trace_packet.inl   k_trace_packet (the camera rays of a dense batch: one wave walks the 4-wide tree for an 8 x 8 pixel block, the node is uniform)
//   rt_trace_wide2.inl    k_trace_wide2 (4-wide top-level tree over 4-wide mesh trees)
//   rt_generate.inl       k_generate (camera rays, slot-per-pixel pipeline)
//   rt_post.inl           post-process, bloom, block errors, texture evaluation
//   rt_kat.inl            known-answer hooks (rtgpu_kat*)
// The host side (rt_runtime.hip) launches them through the declarations of rt_trace_kernels.h.
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#define RT_DEVICE_KERNELS 1
#include "rt_trace_kernels.h"
#include "rt_vcm_state.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

using namespace rtd;

#include "rt_generate.inl"
#include "rt_trace_binary.inl"
#include "rt_wide_grid.inl"
#include "rt_trace_wide.inl"
#include "rt_trace_packet.inl"
#include "rt_trace_wide2.inl"
#include "rt_kat.inl"
#include "rt_post.inl"

// the instantiations the host side launches
#define RT_X(S, C) template __global__ void RT_TRACE_ATTR(S) k_trace<S, C> RT_K_TRACE_ARGS;
RT_K_TRACE_INSTANCES(RT_X)
#undef RT_X
#define RT_X(S, D, L) template __global__ void RT_TRACE_ATTR(S) k_trace_wide<S, D, L> RT_K_TRACE_WIDE_ARGS;
RT_K_TRACE_WIDE_INSTANCES(RT_X)
#undef RT_X
#define RT_X(S) template __global__ void RT_TRACE_ATTR(S) k_trace_wide2<S> RT_K_TRACE_WIDE2_ARGS;
RT_K_TRACE_WIDE2_INSTANCES(RT_X)
#undef RT_X
