// rt_shade.hip -- the shading kernels of the library, a translation unit of their own: PathTracerMIS / PathTracer / Debug shading over
// slot-per-pixel and dense path state (rt_shade.inl, rt_dense.inl) and the bidirectional integrator's kernels (rt_vcm.inl).  The host side
// (rt_runtime.hip) launches them through the declarations of rt_shade_kernels.h.
//
// Why its own unit: it is compiled with -mllvm -simplifycfg-sink-common=false.  SimplifyCFG's common-code sinking merges the stores that
// different branches of the BSDF / shape / counter code make into ONE store through a pointer phi, which SROA cannot promote: the BSDF
// sample record, a pdf and two counters of the generic kernels then live in private (scratch) memory -- 48 ... 192 bytes per lane.  Without
// the sinking: none (generic shade -7 ... -11 %).  The traversal kernels are 0.6 % FASTER with it, hence two units
// (profiles/r03_shade_variants.txt).
#define RT_SHADE_DEFINITIONS 1
#include "rt_device_traverse.h"
#include "rt_vcm_state.h"
#include "rt_shade_kernels.h"

#include "rt_shade.inl"
#include "rt_dense.inl"
#include "rt_vcm.inl"

// the instantiations the host side launches (the lists are in rt_shade_kernels.h)
#define RT_X(L, P, A) template __global__ void RT_SHADE_DENSE_ATTR(L, A) k_shade_dense<L, P, A> RT_K_SHADE_DENSE_ARGS;
RT_K_SHADE_DENSE_INSTANCES(RT_X)
#undef RT_X
#define RT_X(L, P) template __global__ void __launch_bounds__(RT_BLOCK) k_shade<L, P> RT_K_SHADE_ARGS;
RT_K_SHADE_INSTANCES(RT_X)
#undef RT_X
#define RT_X(C) template __global__ void __launch_bounds__(RT_BLOCK) k_vcm_emit<C> RT_K_VCM_EMIT_ARGS; \
                template __global__ void __launch_bounds__(RT_BLOCK) k_vcm_light_shade<C> RT_K_VCM_LIGHT_SHADE_ARGS; \
                template __global__ void __launch_bounds__(RT_BLOCK) k_lt_shade<C> RT_K_LT_SHADE_ARGS; \
                template __global__ void __launch_bounds__(RT_BLOCK) k_vcm_camera_shade<C> RT_K_VCM_CAMERA_SHADE_ARGS;
RT_VCM_CLASSES(RT_X)
#undef RT_X
