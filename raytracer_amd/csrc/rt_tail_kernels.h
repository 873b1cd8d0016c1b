// rt_tail_kernels.h -- the kernel rt_tail.hip defines, declared for the host side (rt_runtime.hip)
#pragma once
#include "rt_trace_common.h"

#define RT_TAIL_PATHS 512u   // path vertices a block of k_tail works on at a time (its five LDS lists hold that many entries each)

struct TailArgs
{
    const uint32_t* denseCounts;   // live / zombie counts of the regions of the arena at the hand-over bounce (DenseCounts::in)
    uint32_t shardCapacity;
    uint32_t* cursor;              // work cursor over the hand-over bounce's vertices (zeroed by the host)
    uint32_t refillMinIdle, otherMinLanes;
    uint32_t* errorFlags;          // RtgpuContext::deviceFlags: [0] = a region of the arena overflowed (checked here as k_shade_dense's prologue does)
    uint32_t anyHitFarFirst;       // WideTuning::anyHitFarFirst for the walks of this launch
};

// X(scene class of rt_device_core.h, plain path tracer)
#define RT_K_TAIL_INSTANCES(X) X(0, false) X(1, false) X(2, false) X(3, false) X(4, false) X(0, true)
#define RT_K_TAIL_ARGS (const RtSceneDesc scene, const WideBvh bvh, const DevPass* __restrict__ passes, uint32_t slotsPerPass, const Paths paths, const TailArgs args, \
                        float4* __restrict__ home, unsigned long long* counters)
// occupancy the register allocator is held to: the walks fit 96 VGPRs, the lean shading code 122 (rt_dense.inl), the generic one 168: four waves per
// SIMD for the lean classes, three for the others (left alone the allocator takes 192 ... 239)
#define RT_TAIL_ATTR(kLean, kPlain) __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu((((kLean) == 1 || (kLean) == 4) && !(kPlain)) ? 4 : 3)))
#ifndef RT_DEVICE_KERNELS
template <int kLean, bool kPlain> __global__ void RT_TAIL_ATTR(kLean, kPlain) k_tail RT_K_TAIL_ARGS;
#endif
