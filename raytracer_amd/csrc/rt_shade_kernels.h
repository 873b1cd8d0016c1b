// rt_shade_kernels.h -- declarations of the kernels rt_shade.hip defines, for the host side in rt_runtime.hip
#pragma once
#include "rt_vcm_state.h"

// one list of k_shade_dense's instantiations -- X(scene class, plain path tracer, LightSamplingStrategy::All) -- for the explicit instantiations in
// rt_shade.hip; the launch ladder of flushBatch (rt_runtime.hip) picks among exactly these
#define RT_SHADE_DENSE_ATTR(kLean, kAll) __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_MIN_WAVES(kLean, kAll), RT_SHADE_MIN_WAVES(kLean, kAll) > 1 ? RT_SHADE_MIN_WAVES(kLean, kAll) : 10)))
#define RT_K_SHADE_DENSE_ARGS (const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths in, const Paths out, const DenseCounts dense, \
                               uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount, float4* __restrict__ home, unsigned long long* counters)
#define RT_K_SHADE_DENSE_INSTANCES(X) X(0, true, false) X(1, false, true) X(2, false, true) X(4, false, true) X(0, false, true) \
                                      X(1, false, false) X(2, false, false) X(3, false, false) X(4, false, false) X(0, false, false)
#define RT_K_SHADE_ARGS (const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths paths, const uint32_t* __restrict__ queueIn, \
                         const uint32_t* __restrict__ countIn, uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut, uint32_t* __restrict__ shadowQueue, \
                         uint32_t* __restrict__ shadowCount, unsigned long long* counters)
#define RT_K_SHADE_INSTANCES(X) X(false, false) X(false, true) X(true, false)


// The bidirectional integrator's kernels that evaluate textures (normal maps, textured material parameters, an environment map) come in two scene
// classes, as k_shade_dense does: 0 = anything, 3 = a scene without textures (rtgpu_upload_scene decides: RtgpuContext::leanScene 1 or 3).
// RT_VCM_WAVES_<kernel>(class): the occupancy the kernel is held to (1 = whatever its registers allow).
#ifndef RT_VCM_WAVES_k_vcm_emit
#define RT_VCM_WAVES_k_vcm_emit(c) 1
#endif
#ifndef RT_VCM_WAVES_k_vcm_light_shade
#define RT_VCM_WAVES_k_vcm_light_shade(c) 1
#endif
#ifndef RT_VCM_WAVES_k_lt_shade
#define RT_VCM_WAVES_k_lt_shade(c) 1
#endif
#ifndef RT_VCM_WAVES_k_vcm_camera_shade
#define RT_VCM_WAVES_k_vcm_camera_shade(c) 1
#endif
#ifndef RT_VCM_CONNECT_WAVES
#define RT_VCM_CONNECT_WAVES 1
#endif
#define RT_VCM_CONNECT_ATTR __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_VCM_CONNECT_WAVES, RT_VCM_CONNECT_WAVES > 1 ? RT_VCM_CONNECT_WAVES : 10)))
#define RT_VCM_ATTR(k) __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_VCM_WAVES_##k(kClass), RT_VCM_WAVES_##k(kClass) > 1 ? RT_VCM_WAVES_##k(kClass) : 10)))
#define RT_VCM_CLASSES(X) X(0) X(3)
#define RT_K_VCM_EMIT_ARGS (const RtSceneDesc scene, const VcmBatch b, const Paths lp, const Paths cp, const VcmArena a, const uint32_t* __restrict__ slotPixel, uint32_t numSlots, uint32_t* __restrict__ queue, uint32_t* __restrict__ queueCount)
#define RT_K_VCM_LIGHT_SHADE_ARGS (const RtSceneDesc scene, const VcmBatch b, const Paths lp, const VcmArena a, const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn, uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount, float* __restrict__ sum, float* __restrict__ secondary, unsigned long long* counters)
#define RT_K_LT_SHADE_ARGS (const RtSceneDesc scene, const VcmBatch b, const Paths lp, const VcmArena a, const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn, uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount, float* __restrict__ sum, float* __restrict__ secondary, unsigned long long* counters)
#define RT_K_VCM_CAMERA_SHADE_ARGS (const RtSceneDesc scene, const VcmBatch b, const Paths cp, const VcmArena a, const uint32_t* __restrict__ queueIn, const uint32_t* __restrict__ countIn, uint32_t* __restrict__ queueOut, uint32_t* __restrict__ countOut, uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount, uint32_t* __restrict__ mergeQueue, uint32_t* __restrict__ mergeCount, uint32_t* __restrict__ connectQueue, uint32_t* __restrict__ connectCount, unsigned long long* counters)

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
template <int kClass>
__global__ void RT_VCM_ATTR(k_vcm_emit) k_vcm_emit RT_K_VCM_EMIT_ARGS;
template <int kClass>
__global__ void RT_VCM_ATTR(k_vcm_light_shade) k_vcm_light_shade RT_K_VCM_LIGHT_SHADE_ARGS;
template <int kClass>
__global__ void RT_VCM_ATTR(k_lt_shade) k_lt_shade RT_K_LT_SHADE_ARGS;
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_light_finish(const VcmBatch b, const Paths lp, uint32_t numSlots, float* __restrict__ sum,
                                                               float* __restrict__ secondary, unsigned long long* counters);
template <int kClass>
__global__ void RT_VCM_ATTR(k_vcm_camera_shade) k_vcm_camera_shade RT_K_VCM_CAMERA_SHADE_ARGS;
__global__ void RT_VCM_CONNECT_ATTR k_vcm_connect(const RtSceneDesc scene, const VcmBatch b, const Paths cp, const VcmArena a,
                                                          const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount,
                                                          uint32_t* __restrict__ shadowQueue, uint32_t* __restrict__ shadowCount);
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_merge(const RtSceneDesc scene, const VcmBatch b, const VcmArena a,
                                                        const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount, uint32_t cooperativeMin);
__global__ void __launch_bounds__(RT_BLOCK) k_vcm_camera_finish(const VcmBatch b, uint32_t numPasses, const Paths cp, const VcmArena a,
                                                                float* __restrict__ sum, float* __restrict__ secondary, uint32_t width, unsigned long long* counters);
#endif   // RT_SHADE_DEFINITIONS
