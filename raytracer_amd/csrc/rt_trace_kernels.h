// rt_trace_kernels.h -- the kernels rt_trace.hip defines (traversal, camera rays, post-process, known-answer hooks), declared for the host side
// (rt_runtime.hip), and the argument structs both sides share.  rt_trace.hip includes it for the structs only (RT_DEVICE_KERNELS).
#pragma once
#include "rt_trace_common.h"

// known-answer hooks (rt_kat.inl): function ids = the ids the fixtures of tests/golden/*.kat carry in their headers (tests/golden/README.md)
enum
{
    KAT_SIN_LANE = 1, KAT_SINCOS = 2, KAT_FASTLOG = 3, KAT_FASTACOS = 4, KAT_FASTATAN2 = 5,
    KAT_FLOAT_NORMAL2 = 6, KAT_HEMISPHERE_COS = 7, KAT_SPHERE = 8, KAT_CIRCLE = 9, KAT_ORTHO_BASIS = 10,
    KAT_FRESNEL_DIELECTRIC = 11, KAT_FRESNEL_METAL = 12, KAT_REFRACT3 = 13, KAT_REFLECT3 = 14,
    KAT_BOX_RAY = 20, KAT_BOX_RAY_TWOSIDED = 21, KAT_TRIANGLE_RAY = 22, KAT_MAKE_RAY = 23, KAT_TRANSFORM_RAY = 24,
    KAT_FAST_INVERSE = 25, KAT_TRANSFORM_SCALED = 26, KAT_FRAME_COMPOSE = 27,
    KAT_SHAPE_INTERSECT = 30, KAT_SHAPE_SAMPLE = 31, KAT_SHAPE_PDF = 32, KAT_SHAPE_EVAL = 33,
    KAT_LIGHT_ILLUMINATE = 40, KAT_LIGHT_RADIANCE = 41, KAT_LIGHT_EMIT = 42, KAT_LIGHT_ILLUMINATE_BIDIR = 43, KAT_LIGHT_RADIANCE_BIDIR = 44,
    KAT_BSDF_SAMPLE = 50, KAT_BSDF_EVALUATE = 51, KAT_BSDF_PDFS = 52,
    KAT_CAMERA_RAY = 60, KAT_CAMERA_FILM = 61, KAT_FILM_SPLAT = 62, KAT_PACKED_PHOTON = 63, KAT_HSV_TO_RGB = 64,
};
#define RT_KAT_MESH_STACK 64

struct PostScale { float c[3]; };
struct BlurPlan { uint32_t n, wl, wu; float m; };
struct BloomLevels { const float* level[5]; };
struct ErrorRow { uint32_t block, y; };

// one list of the traversal kernels' instantiations: X(stack entries per lane, intersection counters) etc.
#define RT_TRACE_ATTR(kStack) __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(kStack <= 24 ? 5 : 1)))
#define RT_K_TRACE_ARGS (const RtSceneDesc scene, const Paths paths, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount, \
                         const uint32_t* __restrict__ shadowQueue, const uint32_t* __restrict__ shadowCount, uint32_t* __restrict__ cursor, unsigned long long* counters, const TravTuning tune)
#define RT_K_TRACE_WIDE_ARGS (const RtSceneDesc scene, const WideBvh bvh, const Paths paths, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount, \
                              const uint32_t* __restrict__ shadowQueue, const uint32_t* __restrict__ shadowCount, uint32_t* __restrict__ cursor, unsigned long long* counters, const WideTuning tune)
#define RT_K_TRACE_WIDE2_ARGS (const RtSceneDesc scene, const WideScene wide, const Paths paths, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount, \
                               const uint32_t* __restrict__ shadowQueue, const uint32_t* __restrict__ shadowCount, uint32_t* __restrict__ cursor, unsigned long long* counters, const WideTuning tune)
#define RT_K_TRACE_INSTANCES(X) X(24, false) X(24, true) X(32, false) X(32, true) X(64, false) X(64, true)
#define RT_K_TRACE_WIDE_INSTANCES(X) X(24, false, false) X(24, false, true) X(24, true, false)
#define RT_K_TRACE_WIDE2_INSTANCES(X) X(24)

#ifndef RT_DEVICE_KERNELS
template <int kStack, bool kCount> __global__ void RT_TRACE_ATTR(kStack) k_trace RT_K_TRACE_ARGS;
template <int kStack, bool kDiag = false, bool kLocalExact = false> __global__ void RT_TRACE_ATTR(kStack) k_trace_wide RT_K_TRACE_WIDE_ARGS;
template <int kStack> __global__ void RT_TRACE_ATTR(kStack) k_trace_wide2 RT_K_TRACE_WIDE2_ARGS;
__global__ void __launch_bounds__(RT_BLOCK) k_trace_packet(const RtSceneDesc scene, const WideBvh bvh, const Paths paths, uint32_t* __restrict__ cursor, unsigned long long* counters, const WideTuning tune);
__global__ void __launch_bounds__(RT_BLOCK) k_generate(const RtSceneDesc scene, const DevPass* __restrict__ passes, uint32_t slotsPerPass,
                                                       const Paths paths, const uint32_t* __restrict__ slotPixel, uint32_t numSlots,
                                                       uint32_t* __restrict__ queue, uint32_t* __restrict__ queueCount,
                                                       unsigned long long* counters);
__global__ void __launch_bounds__(RT_BLOCK) k_postprocess(const float* __restrict__ sum, uint32_t* __restrict__ front, uint32_t width, uint32_t height,
                                                          const RtPostprocessParams params, const PostScale colorScale);
__global__ void __launch_bounds__(RT_BLOCK) k_blur_lines(float* __restrict__ image, uint32_t width, uint32_t height, uint32_t vertical, const BlurPlan plan,
                                                         float* __restrict__ lineA, float* __restrict__ lineB);
__global__ void __launch_bounds__(RT_BLOCK) k_postprocess_bloom(const float* __restrict__ sum, const BloomLevels blurred, uint32_t* __restrict__ front, uint32_t width, uint32_t height,
                                                                const RtPostprocessParams params, const PostScale colorScale);
__global__ void __launch_bounds__(RT_BLOCK) k_block_error_rows(const float* __restrict__ sum, const float* __restrict__ secondary, uint32_t width,
                                                               const RtBlock* __restrict__ blocks, const ErrorRow* __restrict__ rows, uint32_t numRows,
                                                               float imageScalingFactor, float* __restrict__ rowErrors);
__global__ void __launch_bounds__(RT_BLOCK) k_block_error_total(const RtBlock* __restrict__ blocks, const uint32_t* __restrict__ firstRow, uint32_t numBlocks,
                                                                const float* __restrict__ rowErrors, uint32_t totalArea, float* __restrict__ outErrors);
__global__ void __launch_bounds__(RT_BLOCK) k_evaluate_textures(const RtSceneDesc scene, uint32_t count, const uint32_t* __restrict__ textureIndex,
                                                                const float* __restrict__ uv, float* __restrict__ out);
__global__ void __launch_bounds__(64) k_kat(const RtSceneDesc scene, uint32_t func, const float* __restrict__ in, uint32_t inStride, float* __restrict__ out,
                                            uint32_t outStride, uint32_t n);
__global__ void __launch_bounds__(64) k_kat_sampler(const uint16_t* __restrict__ blueNoise, const float* __restrict__ in, uint32_t inStride, float* __restrict__ out,
                                                    uint32_t count, uint32_t n);
__global__ void __launch_bounds__(64) k_kat_mesh(const RtSceneDesc scene, const float* __restrict__ rays, uint32_t n, uint32_t* __restrict__ out);
__global__ void __launch_bounds__(RT_MONSTER_BLOCK) k_trace_monster(const RtSceneDesc scene, const Paths paths, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ queueCount);
#endif
