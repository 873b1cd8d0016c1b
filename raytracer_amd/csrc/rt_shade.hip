// rt_shade.hip -- the shading kernels of the library, a translation unit of their own: PathTracerMIS / PathTracer / Debug shading over
// slot-per-pixel and dense path state (rt_shade.inl, rt_dense.inl) and the bidirectional integrator's kernels (rt_vcm.inl).  The host side
// (rt_runtime.hip) launches them through the declarations of rt_shade_kernels.h.
//
// Why its own unit: it is compiled with -mllvm -simplifycfg-sink-common=false.  SimplifyCFG's common-code sinking merges the stores that
// different branches of the BSDF / shape / counter code make into ONE store through a pointer phi, which SROA cannot promote: the BSDF
// sample record, a pdf and two counters of the generic kernels then live in private (scratch) memory -- 48 ... 192 bytes per lane.  Without
// the sinking: none (generic shade -7 ... -11 %).  The traversal kernels are 0.6 % FASTER with it, hence two units
// (profiles/r03_shade_variants.txt).
#include "rt_device_traverse.h"
#include "rt_vcm_state.h"

#include "rt_shade.inl"
#include "rt_dense.inl"
#include "rt_vcm.inl"

// the instantiations the host side launches
template __global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(0, false), RT_SHADE_MIN_WAVES(0, false) > 1 ? RT_SHADE_MIN_WAVES(0, false) : 10))) k_shade_dense<0, true, false>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds);
template __global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(1, true), RT_SHADE_MIN_WAVES(1, true) > 1 ? RT_SHADE_MIN_WAVES(1, true) : 10))) k_shade_dense<1, false, true>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds);
template __global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(2, true), RT_SHADE_MIN_WAVES(2, true) > 1 ? RT_SHADE_MIN_WAVES(2, true) : 10))) k_shade_dense<2, false, true>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds);
template __global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(4, true), RT_SHADE_MIN_WAVES(4, true) > 1 ? RT_SHADE_MIN_WAVES(4, true) : 10))) k_shade_dense<4, false, true>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds);
template __global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(0, true), RT_SHADE_MIN_WAVES(0, true) > 1 ? RT_SHADE_MIN_WAVES(0, true) : 10))) k_shade_dense<0, false, true>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds);
template __global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(1, false), RT_SHADE_MIN_WAVES(1, false) > 1 ? RT_SHADE_MIN_WAVES(1, false) : 10))) k_shade_dense<1, false, false>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds);
template __global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(2, false), RT_SHADE_MIN_WAVES(2, false) > 1 ? RT_SHADE_MIN_WAVES(2, false) : 10))) k_shade_dense<2, false, false>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds);
template __global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(3, false), RT_SHADE_MIN_WAVES(3, false) > 1 ? RT_SHADE_MIN_WAVES(3, false) : 10))) k_shade_dense<3, false, false>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds);
template __global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(4, false), RT_SHADE_MIN_WAVES(4, false) > 1 ? RT_SHADE_MIN_WAVES(4, false) : 10))) k_shade_dense<4, false, false>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds);
template __global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(0, false), RT_SHADE_MIN_WAVES(0, false) > 1 ? RT_SHADE_MIN_WAVES(0, false) : 10))) k_shade_dense<0, false, false>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out,
                                                          const DenseCounts dense, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                          float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds);
template __global__ void __launch_bounds__(RT_BLOCK) k_shade<false, false>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths paths,
                                                    const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                    uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                    uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                    unsigned long long* counters);
template __global__ void __launch_bounds__(RT_BLOCK) k_shade<false, true>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths paths,
                                                    const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                    uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                    uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                    unsigned long long* counters);
template __global__ void __launch_bounds__(RT_BLOCK) k_shade<true, false>(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths paths,
                                                    const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                    uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                    uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                    unsigned long long* counters);
