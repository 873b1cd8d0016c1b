// rt_shade_kernels.h -- declarations of the kernels rt_shade.hip defines, for the host side in rt_runtime.hip
#pragma once
#include "rt_vcm_state.h"

// one list of k_shade_dense's instantiations -- X(scene class, plain path tracer, LightSamplingStrategy::All) -- for the explicit instantiations in
// rt_shade.hip; the launch ladder of flushBatch (rt_runtime.hip) picks among exactly these
#define RT_SHADE_DENSE_ATTR(kLean, kAll) __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(kLean, kAll), RT_SHADE_MIN_WAVES(kLean, kAll) > 1 ? RT_SHADE_MIN_WAVES(kLean, kAll) : 10)))
#define RT_K_SHADE_DENSE_ARGS (const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out, const DenseCounts dense, \
                               uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount, float4* __restrict__ home, unsigned long long* counters, uint32_t sortKinds)
#define RT_K_SHADE_DENSE_INSTANCES(X) X(0, true, false) X(1, false, true) X(2, false, true) X(4, false, true) X(0, false, true) \
                                      X(1, false, false) X(2, false, false) X(3, false, false) X(4, false, false) X(0, false, false)
#define RT_K_SHADE_ARGS (const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths paths, const uint32_t* __restrict__ queueIn, \
                         const uint32_t* __restrict__ countIn, uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut, uint32_t* __restrict__ shadowQueue, \
                         uint32_t* __restrict__ shadowCount, unsigned long long* counters)
#define RT_K_SHADE_INSTANCES(X) X(false, false) X(false, true) X(true, false)

#ifndef RT_SHADE_DEFINITIONS   // (rt_shade.hip and rt_tail.hip define the kernels: they take the lists above only)
template <bool kLean, bool kPlain = false>
__global__ void __launch_bounds__(RT_BLOCK) k_shade(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths paths,
                                                    const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                    uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                    uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                    unsigned long long* counters);
__global__ void __launch_bounds__(RT_BLOCK) k_debug_shade(const RtSceneDesc scene, const Paths paths, const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                          uint32_t mode, unsigned long long* counters);
__global__ void __launch_bounds__(RT_BLOCK) k_accumulate(const Paths paths, uint32_t slotsPerPass, uint32_t numPasses, float* __restrict__ sum,
                                                         float* __restrict__ secondary, uint32_t width, const DevPass* __restrict__ passes,
                                                         unsigned long long* counters);
__global__ void __launch_bounds__(RT_BLOCK) k_generate_dense(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths paths,
                                                             const uint32_t* __restrict__ slotPixel, uint32_t numSlots, uint32_t shardCapacity, uint32_t* __restrict__ counts,
                                                             unsigned long long* counters, uint32_t fullRecords);
template <int kLean, bool kPlain = false, bool kAll = false>
__global__ void RT_SHADE_DENSE_ATTR(kLean, kAll) k_shade_dense RT_K_SHADE_DENSE_ARGS;
__global__ void __launch_bounds__(RT_BLOCK) k_accumulate_home(const float4* __restrict__ home, const uint32_t* __restrict__ slotPixel, uint32_t slotsPerPass, uint32_t numPasses,
                                                              float* __restrict__ sum, float* __restrict__ secondary, uint32_t width, const DevPass* __restrict__ passes);
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_emit(const RtSceneDesc scene, const VcmBatch b, const Paths lp, const Paths cp,
                                                       const VcmArena a, const uint32_t* __restrict__ slotPixel, uint32_t numSlots,
                                                       uint32_t* __restrict__ queue, uint32_t* __restrict__ queueCount);
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_light_shade(const RtSceneDesc scene, const VcmBatch b, const Paths lp, const VcmArena a,
                                                              const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                              uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                              uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                              float* __restrict__ sum, float* __restrict__ secondary, unsigned long long* counters);
__global__ void __launch_bounds__(RT_BLOCK) k_lt_shade(const RtSceneDesc scene, const VcmBatch b, const Paths lp, const VcmArena a,
                                                       const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                       uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                       uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                       float* __restrict__ sum, float* __restrict__ secondary, unsigned long long* counters);
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_light_finish(const VcmBatch b, const Paths lp, uint32_t numSlots, float* __restrict__ sum,
                                                               float* __restrict__ secondary, unsigned long long* counters);
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_camera_shade(const RtSceneDesc scene, const VcmBatch b, const Paths cp, const VcmArena a,
                                                               const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn,
                                                               uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut,
                                                               uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount,
                                                               uint32_t* __restrict__ mergeQueue, uint32_t* __restrict__ mergeCount,
                                                               uint32_t* __restrict__ connectQueue, uint32_t* __restrict__ connectCount, unsigned long long* counters);
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_connect(const RtSceneDesc scene, const VcmBatch b, const Paths cp, const VcmArena a,
                                                          const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount,
                                                          uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount);
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_merge(const RtSceneDesc scene, const VcmBatch b, const VcmArena a,
                                                        const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount, uint32_t cooperativeMin);
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_camera_finish(const VcmBatch b, uint32_t numPasses, const Paths cp, const VcmArena a,
                                                                float* __restrict__ sum, float* __restrict__ secondary, uint32_t width, unsigned long long* counters);
#endif   // RT_SHADE_DEFINITIONS
