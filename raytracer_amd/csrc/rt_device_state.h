// rt_device_state.h -- device-side path state shared by the translation units of the library (rt_runtime.hip: host side; rt_trace.hip: traversal;
// rt_shade.hip: the shading kernels): the record arenas, per-pass constants, counter / sampler plumbing and the dense-arena helpers.
#pragma once
#include "rt_device_core.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
using namespace rtd;

// =====================================================================================================
// Device-side data
// =====================================================================================================
// Path state lives in HBM as 16-byte RECORDS, one array per record kind (record-major, slot-minor): a lane moves a
// whole record with one dwordx4 access, a wave's accesses to consecutive slots coalesce into 1 KB, and a path
// vertex touches 7-9 arrays (and as many DRAM pages / TLB entries) instead of 34 scalar planes.
enum PathRecord : uint32_t
{
    R_ORIGIN,    // ray origin xyz BEFORE the 1e-3 offset | flags: depth (bits 0-7), lastSpecular << 8, (previous vertex's material + 1) << 9
    R_DIR,       // ray direction xyz as passed to Ray()   | lastPdfW
    R_TP,        // throughput (4 lanes: RayColor::AlmostZero tests all four)
    R_RESULT,    // accumulated radiance rgb of this path  | pixel: x | y << 16
    R_HIT,       // objectId, subObjectId, distance, u
    R_SAMPLER,   // hit v | GenericSampler salt, generated | number of NEE requests pending for this vertex
    R_RNG,       // per-pixel xoroshiro128+ state (2 x 64 bit)
    R_SH_P,      // shading point xyz (shadow ray origin before the 1e-4 offset)
    R_SH_TP,     // throughput at the vertex (the NEE fma uses it)
    R_NUM_BASE
};
// per NEE request two records, light-major: {direction xyz, tmax (< 0: no ray / occluded)}, {contribution rgb, -}
#define RT_SHADOW_RECORDS 2

struct Paths
{
    float4* base;       // (R_NUM_BASE + maxLights * RT_SHADOW_RECORDS) * capacity records
    uint32_t capacity;
    uint32_t maxLights; // NEE requests per vertex (1 for LightSamplingStrategy::Single)
};

RT_DEV float4& prec(const Paths& p, uint32_t record, uint32_t slot) { return p.base[(size_t)record * p.capacity + slot]; }
RT_DEV float4& pshadow(const Paths& p, uint32_t light, uint32_t k, uint32_t slot)
{
    return p.base[((size_t)R_NUM_BASE + (size_t)light * RT_SHADOW_RECORDS + k) * p.capacity + slot];
}
RT_DEV float4 f4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
// Path records are STREAMED: a bounce reads a record once and writes its successor once, gigabytes per launch, through the same 4 MB-per-XCD
// L2 that holds the scene's nodes, triangles and shading records.  Non-temporal accesses keep the stream from evicting the geometry.
#ifndef RT_STREAMING_HINTS
#define RT_STREAMING_HINTS 1
#endif
typedef float rt_float4v __attribute__((ext_vector_type(4)));
RT_DEV float4 ldStream(const float4& r)
{
#if RT_STREAMING_HINTS
    const rt_float4v v = __builtin_nontemporal_load(reinterpret_cast<const rt_float4v*>(&r));
    return f4(v.x, v.y, v.z, v.w);
#else
    return r;
#endif
}
RT_DEV void stStream(float4& r, const float4& v)
{
#if RT_STREAMING_HINTS
    const rt_float4v t = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(t, reinterpret_cast<rt_float4v*>(&r));
#else
    r = v;
#endif
}
RT_DEV float fbits(uint32_t u) { return __uint_as_float(u); }
RT_DEV uint32_t ubits(float f) { return __float_as_uint(f); }

struct DevPass
{
    RtCamera camera;
    const uint32_t* seed;
    uint32_t numDimensions;
    uint32_t blueNoiseLayers;
    float sampleOffset[2];
    uint32_t passIndex;
    uint32_t maxRayDepth;
    uint32_t minRussianRouletteDepth;
    uint32_t lightSamplingStrategy;
    float lightSamplingWeight[4];
    float bsdfSamplingWeight[4];
    uint64_t rngKey[2];
    uint32_t width, height;
};

#define RT_BLOCK 256

// per-block counter flush: LDS tally, then one 64-bit atomic per counter per block
RT_DEV void flushCounters(const Counters& c, unsigned long long* global)
{
    __shared__ uint32_t sC[RT_NUM_COUNTERS];
    if (threadIdx.x < RT_NUM_COUNTERS) sC[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RT_NUM_COUNTERS; ++k) if (c.c[k]) atomicAdd(&sC[k], c.c[k]);
    __syncthreads();
    if (threadIdx.x < RT_NUM_COUNTERS && sC[threadIdx.x]) atomicAdd(&global[threadIdx.x], (unsigned long long)sC[threadIdx.x]);
}
RT_DEV void zeroCounters(Counters& c) {
#pragma unroll
    for (int k = 0; k < RT_NUM_COUNTERS; ++k) c.c[k] = 0;
}

// GenericSampler + per-pixel RNG state of a path (R_SAMPLER.yz, R_RNG); `sampler` is the record already loaded
RT_DEV void loadSampler(Sampler& s, const Paths& p, uint32_t slot, uint32_t pix, const float4& sampler, const DevPass& pass, const uint16_t* blueNoise)
{
    s.seed = pass.seed; s.numDims = pass.numDimensions; s.blueNoiseLayers = pass.blueNoiseLayers; s.blueNoise = blueNoise;
    s.bx = (pix & 0xFFFFu) & 127u; s.by = (pix >> 16) & 127u;
    s.salt = ubits(sampler.y); s.generated = ubits(sampler.z);
    const float4 rng = prec(p, R_RNG, slot);
    s.fallback.s[0] = (uint64_t)ubits(rng.x) | ((uint64_t)ubits(rng.y) << 32);
    s.fallback.s[1] = (uint64_t)ubits(rng.z) | ((uint64_t)ubits(rng.w) << 32);
}
RT_DEV void storeSampler(const Sampler& s, const Paths& p, uint32_t slot, float hitV, uint32_t pendingRequests)
{
    prec(p, R_SAMPLER, slot) = f4(hitV, fbits(s.salt), fbits(s.generated), fbits(pendingRequests));
    prec(p, R_RNG, slot) = f4(fbits((uint32_t)s.fallback.s[0]), fbits((uint32_t)(s.fallback.s[0] >> 32)),
                              fbits((uint32_t)s.fallback.s[1]), fbits((uint32_t)(s.fallback.s[1] >> 32)));
}

// The path's current ray exactly as the reference holds it: Ray(origin, direction) -- which normalises and
// computes invDir / originDivDir from the UN-offset origin -- and then origin += dir * 0.001f for
// secondary rays, leaving originDivDir stale (PathTracerMIS.cpp:392-393).
RT_DEV Ray makePathRay(const float4& origin, const float4& dir, uint32_t depth)
{
    Ray ray = makeRay(V4(origin.x, origin.y, origin.z, 0.0f), V4(dir.x, dir.y, dir.z, 0.0f));
    if (depth > 0) ray.origin = ray.origin + ray.dir * 0.001f;
    return ray;
}

// ---- dense path state (rt_dense.inl): an arena is RT_DENSE_SHARDS regions, region s holds its live paths upwards from s * shardCapacity ----
#define RT_DENSE_SHARDS 16u

struct DenseCounts
{
    const uint32_t* in;     // [0, 16): live paths per region of the arena being read; [16, 32): zombies per region
    uint32_t* out;          // the same for the arena being written (zeroed by the host)
    uint32_t shardCapacity;
    uint32_t* errorFlags;   // host-visible words of the context (RtgpuContext::deviceFlags): [0] != 0 = a region of the arena overflowed
    const uint32_t* primarySlotPixel;   // bounce 0 of a batch whose k_generate_dense stored origin and direction only: the slot -> pixel table; else null
};

// prefix sums of the 16 region counts into LDS (prefix[16] = total); all threads of the block call it
RT_DEV void denseLoadPrefix(const uint32_t* __restrict__ counts, uint32_t* sPrefix)
{
    if (threadIdx.x == 0)
    {
        uint32_t sum = 0;
        for (uint32_t s = 0; s < RT_DENSE_SHARDS; ++s) { sPrefix[s] = sum; sum += counts[s]; }
        sPrefix[RT_DENSE_SHARDS] = sum;
    }
}
RT_DEV uint32_t denseRegionOf(const uint32_t* sPrefix, uint32_t idx)
{
    uint32_t s = idx >= sPrefix[8] ? 8u : 0u;
    s += idx >= sPrefix[s + 4u] ? 4u : 0u;
    s += idx >= sPrefix[s + 2u] ? 2u : 0u;
    s += idx >= sPrefix[s + 1u] ? 1u : 0u;
    return s;
}
// slot of the idx-th live path
RT_DEV uint32_t denseLiveSlot(const uint32_t* sPrefix, uint32_t shardCapacity, uint32_t idx)
{
    const uint32_t s = denseRegionOf(sPrefix, idx);
    return s * shardCapacity + (idx - sPrefix[s]);
}

#define RT_DENSE_MAX_LIGHTS 7u   // 256 vertices x 7 requests fit the block's append buffer between two flushes (k_shade_dense)

// Occupancy the register allocator is held to per scene class: "lean + simple bitmaps" needs 135 VGPRs on its own and fits four waves per SIMD with
// 8-16 bytes of scratch (measured +4 % end to end on the textured Sponza-class scene); "lean + textures" 173 -> 168 = three waves (+8 %), "anything"
// 192 -> 168 = three waves (+1.5 %); the lean and the untextured classes keep what they get (forcing THEM further was slower,
// profiles/r03_shade_variants.txt).
#define RT_SHADE_MIN_WAVES(k, all) ((k) == 4 ? 4 : (((k) == 2 || ((k) == 0 && !(all))) ? 3 : 1))   // ("anything" under `All`: 216 VGPRs, left alone)

// ---- DebugRenderer (renderer name "Debug"): modes and the TriangleID colour (also a known-answer function of rt_kat.inl) ----
// DebugRenderer::RenderPixel after the primary ray's traversal (Core/Rendering/DebugRenderer.cpp:26-195, renderer "Debug"): one colour
// per pixel from the first hit.  mode = DebugRenderingMode (DebugRenderer.h:7-33; the four counter modes exist only under
// RT_ENABLE_INTERSECTION_COUNTERS, off in the reference).
enum { DBG_CAMERA_LIGHT = 0, DBG_TRIANGLE_ID, DBG_DEPTH, DBG_POSITION, DBG_NORMALS, DBG_TANGENTS, DBG_BITANGENTS, DBG_TEXCOORDS,
       DBG_BASE_COLOR, DBG_EMISSION, DBG_ROUGHNESS, DBG_METALNESS, DBG_IOR, DBG_NUM_MODES };
RT_DEV V4 hsvToRgb(float hue, float saturation, float value)   // Core/Color/ColorHelpers.h:133-156
{
    const int h_i = (int)(hue * 6.0f);
    const float f = hue * 6 - h_i;
    const float p = value * (1 - saturation);
    const float q = value * (1 - f * saturation);
    const float t = value * (1 - (1 - f) * saturation);
    if (h_i == 0) return V4(value, t, p, 0.0f);
    else if (h_i == 1) return V4(q, value, p, 0.0f);
    else if (h_i == 2) return V4(p, value, t, 0.0f);
    else if (h_i == 3) return V4(p, q, value, 0.0f);
    else if (h_i == 4) return V4(t, p, value, 0.0f);
    else if (h_i == 5) return V4(value, p, q, 0.0f);
    return zero4();
}
RT_DEV V4 debugTriangleIdColor(uint32_t objectId, uint32_t subObjectId)   // DebugRenderer.cpp:98-106
{
    const uint64_t hash = murmurFmix64((uint64_t)objectId | ((uint64_t)subObjectId << 32));
    const float hue = (float)(uint32_t)hash / (float)UINT32_MAX;
    const float saturation = 0.5f + 0.5f * (float)(uint32_t)(hash >> 32) / (float)UINT32_MAX;
    return hsvToRgb(hue, saturation, 1.0f);
}
