// rt_tail.hip -- the FUSED TAIL of a batch: one persistent launch that takes every path alive at a hand-over bounce through the rest of
// PathTracerMIS::RenderPixel's loop (PathTracerMIS.cpp:270-396) -- trace, the reference's own walk for the rays the 4-wide walk does not decide,
// shade, next-event shadow rays -- until the last one has parked its radiance.  Single-mesh scenes (the 4-wide tree of rt_trace_wide.inl), dense
// path state, LightSamplingStrategy::Single or the plain path tracer.
//
// Why.  A batch's bounces are launch triples (trace, re-trace, shade) with a barrier between them, and every launch lasts as long as its longest
// ray: the ten k_trace_wide launches of a 5-pass batch take 1420 / 3930 / 2768 / 1773 / 1093 / 751 / 574 / 457 / 359 / 231 us
// (profiles/r03_timeline_serial_start_of_round.txt) -- the last five bounces hold 11 % of the vertices and cost 18 % of the trace time, 23 % of the
// re-trace time, with a floor of ~0.35 ms per bounce that does not shrink when the frame is sharded over eight devices.  Here the barrier is per
// BLOCK: a block claims RT_TAIL_PATHS vertices of the hand-over bounce and runs its own little wavefront pipeline over them, in place, with its
// queues in LDS --
//     T  traceWideLoop over {closest-hit rays of the block's live paths} + {the pending next-event requests}       (rt_trace_wide.inl)
//     X  traceBinaryLoop over the few rays T did not decide                                                        (rt_trace_binary.inl)
//     S  denseShadeVertex for every live path and every zombie; survivors stay in their slot                       (rt_dense.inl)
// -- until its paths are gone, then claims the next chunk.  Blocks are out of step with each other, so one block's drain overlaps the others'
// bulk; there is no launch boundary, no re-trace launch, no compaction (at this population the records' locality no longer pays for it).
// The arithmetic per path is that of the wavefront kernels (same functions, same order): images and counters are bit-identical whatever the
// hand-over bounce (tests/test_gpu_parity.py).
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -simplifycfg-sink-common=false   (the shading code, see rt_shade.hip)
#define RT_DEVICE_KERNELS 1
#define RT_SHADE_FUNCTIONS_ONLY 1
#define RT_TRACE_FUNCTIONS_ONLY 1
#include "rt_trace_kernels.h"
#include "rt_tail_kernels.h"
#define RT_SHADE_DEFINITIONS 1
#include "rt_shade_kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

using namespace rtd;

#include "rt_trace_binary.inl"
#include "rt_wide_grid.inl"
#include "rt_trace_wide.inl"
#include "rt_shade.inl"
#include "rt_dense.inl"

// Invariant the hand-over lists rest on: a chunk holds RT_TAIL_PATHS vertices, each with at most ONE closest-hit ray and ONE next-event request
// (LightSamplingStrategy::Single or none), so a walk never hands over more than RT_TAIL_PATHS entries of a kind and widePushExact never needs the
// launch-wide queues, which this kernel does not have (WideTuning's queue pointers are null; widePushExact drops the request and raises nothing
// if that ever changed -- kAll in the tail or a larger chunk needs queues of its own).
// LDS of a block: the walks' stacks (24 entries x 256 lanes = 24 KB; the shade phase stages its next-event records in the same memory), five
// lists of RT_TAIL_PATHS slots (live / zombie vertices of this round and the next, shadow requests; the lists of the NEXT round double as the
// hand-over lists of the walks, which run when those are empty) -- 35 KB, four blocks per CU.
enum { TN_LIVE = 0, TN_ZOMBIES, TN_SHADOW, TN_NEXT_LIVE, TN_NEXT_ZOMBIES, TN_CURSOR, TN_BASE, TN_COUNT };

template <int kLean, bool kPlain>
__global__ void RT_TAIL_ATTR(kLean, kPlain) k_tail RT_K_TAIL_ARGS
{
    __shared__ __attribute__((aligned(16))) uint32_t sStack[24 * RT_BLOCK];   // (the 4-wide walk uses RT_WIDE_STACK entries + RT_WIDE_PARK parked words of it, the reference's walk all 24)
    static_assert(RT_WIDE_STACK + (int)RT_WIDE_PARK <= 24, "the 4-wide walk's stack and parked words share the 24-entry stack memory");
    __shared__ uint32_t sLists[5][RT_TAIL_PATHS];
    __shared__ uint32_t sLivePrefix[RT_DENSE_SHARDS + 1u], sZombiePrefix[RT_DENSE_SHARDS + 1u];
    __shared__ uint32_t sN[TN_COUNT];
    float4 (*stage)[RT_BLOCK] = reinterpret_cast<float4 (*)[RT_BLOCK]>(sStack);

    denseLoadPrefix(args.denseCounts, sLivePrefix);
    if (threadIdx.x == 64)
    {
        uint32_t sum = 0;
        for (uint32_t s = 0; s < RT_DENSE_SHARDS; ++s)
        {
            sZombiePrefix[s] = sum; sum += args.denseCounts[RT_DENSE_SHARDS + s];
            // the launch before this one overfilled region s (live paths growing up met the zombies growing down): k_shade_dense's prologue check, which the
            // hand-over bounce would otherwise lose
            if (blockIdx.x == 0 && args.denseCounts[s] + args.denseCounts[RT_DENSE_SHARDS + s] > args.shardCapacity) args.errorFlags[0] = 1u;
        }
        sZombiePrefix[RT_DENSE_SHARDS] = sum;
    }
    __syncthreads();
    const uint32_t numLive = sLivePrefix[RT_DENSE_SHARDS], total = numLive + sZombiePrefix[RT_DENSE_SHARDS];
    // the i-th vertex of the hand-over bounce: live paths first (region by region), then the zombies (from the top of their regions), as k_shade_dense
    auto vertexSlot = [&](uint32_t idx, bool& zombie) -> uint32_t
    {
        zombie = idx >= numLive;
        if (!zombie) return denseLiveSlot(sLivePrefix, args.shardCapacity, idx);
        const uint32_t z = idx - numLive, s = denseRegionOf(sZombiePrefix, z);
        return (s + 1u) * args.shardCapacity - 1u - (z - sZombiePrefix[s]);
    };
    Counters cnt; zeroCounters(cnt);
    const DevPass pass = passes[0];   // the structural parameters are those of every pass of the batch
    const V4 lightSamplingWeight = load4(pass.lightSamplingWeight), bsdfSamplingWeight = load4(pass.bsdfSamplingWeight);
    const float lightPickProbability = 1.0f / (float)(scene.numLights ? scene.numLights : 1u);   // GetLightPickingProbability, PathTracerMIS.cpp:157-172 (Single)
    const WideTuning wideTune = { args.refillMinIdle, args.otherMinLanes, 0.0001f, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 64u, 1u, 0u, 0u, args.anyHitFarFirst };
    const TravTuning exactTune = { args.refillMinIdle, args.otherMinLanes, 0.0001f, nullptr, nullptr, RT_ABORT_CLOSEST_AFTER, nullptr, 0u, RT_RETRACE_SPLIT_AFTER };
    const uint32_t blockWaves = (uint32_t)RT_BLOCK / 64u;
    uint32_t cur = 0u;   // which pair of (live, zombie) lists is this round's; block-uniform

    for (;;)
    {
        // ---- the block's next chunk of the hand-over bounce ----
        if (threadIdx.x < TN_BASE) sN[threadIdx.x] = 0u;
        if (threadIdx.x == 64) sN[TN_BASE] = atomicAdd(args.cursor, RT_TAIL_PATHS);
        __syncthreads();
        const uint32_t base = sN[TN_BASE];
        if (base >= total) break;
        for (uint32_t k = threadIdx.x; k < RT_TAIL_PATHS; k += RT_BLOCK)
        {
            const uint32_t idx = base + k;
            if (idx >= total) continue;
            bool zombie;
            const uint32_t slot = vertexSlot(idx, zombie);
            if (zombie) sLists[2u + cur][atomicAdd(&sN[TN_ZOMBIES], 1u)] = slot; else sLists[cur][atomicAdd(&sN[TN_LIVE], 1u)] = slot;
            // the next-event request the previous bounce's shade left with the vertex: its shadow ray has not been traced yet (request index = light 0 * capacity + slot)
            if (ubits(prec(paths, R_SAMPLER, slot).w) != 0u && pshadow(paths, 0, 0, slot).w >= 0.0f) sLists[4][atomicAdd(&sN[TN_SHADOW], 1u)] = slot;
        }
        __syncthreads();
        while (sN[TN_LIVE] + sN[TN_ZOMBIES] != 0u)
        {
            uint32_t* const live = sLists[cur]; uint32_t* const zombies = sLists[2u + cur];
            uint32_t* const nextLive = sLists[cur ^ 1u]; uint32_t* const nextZombies = sLists[2u + (cur ^ 1u)];
            if (sN[TN_LIVE] + sN[TN_SHADOW] != 0u)
            {
                // ---- T: the 4-wide walk; what it does not decide waits in the (still empty) lists of the next round ----
                const WideLocal handOver = { nextLive, &sN[TN_NEXT_LIVE], nextZombies, &sN[TN_NEXT_ZOMBIES], RT_TAIL_PATHS };
                traceWideLoop<RT_WIDE_STACK, false>(scene, bvh, paths, live, &sN[TN_LIVE], sLists[4], &sN[TN_SHADOW], &sN[TN_CURSOR], counters, wideTune, handOver, sStack, sLivePrefix, blockWaves);
                __syncthreads();
                if (sN[TN_NEXT_LIVE] + sN[TN_NEXT_ZOMBIES] != 0u)
                {
                    // ---- X: the reference's own walk for those ----
                    if (threadIdx.x == 0) sN[TN_CURSOR] = 0u;
                    __syncthreads();
                    traceBinaryLoop<24, false>(scene, paths, nextLive, &sN[TN_NEXT_LIVE], nextZombies, &sN[TN_NEXT_ZOMBIES], &sN[TN_CURSOR], counters, exactTune, sStack, sLivePrefix, blockWaves);
                    __syncthreads();
                }
            }
            // ---- S: one vertex per thread, in place ----
            const uint32_t nLive = sN[TN_LIVE], nAll = nLive + sN[TN_ZOMBIES];
            __syncthreads();
            if (threadIdx.x == 0) { sN[TN_SHADOW] = 0u; sN[TN_NEXT_LIVE] = 0u; sN[TN_NEXT_ZOMBIES] = 0u; sN[TN_CURSOR] = 0u; }
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < nAll; k += RT_BLOCK)
            {
                const bool zombie = k >= nLive;
                const uint32_t slot = zombie ? zombies[k - nLive] : live[k];
                DenseVertex v;
                v.outcome = 0; v.stagedShTp = false; v.rayNeeded = false; v.oHome = 0u; v.rayMask = 0u;
                denseShadeVertex<kLean, kPlain, false>(scene, passes, slotsPerPass, pass, paths, slot, zombie, lightSamplingWeight, bsdfSamplingWeight, lightPickProbability, stage, home, cnt, v);
                if (v.outcome == 0) continue;   // its radiance is parked
                prec(paths, R_RESULT, slot) = v.oResult;
                prec(paths, R_SAMPLER, slot) = v.oSampler;
                prec(paths, R_SH_TP, slot) = v.stagedShTp ? stage[3][threadIdx.x] : f4(0.0f, 0.0f, 0.0f, fbits(v.oHome));
                if (v.outcome == 1)
                {
                    prec(paths, R_ORIGIN, slot) = v.oOrigin; prec(paths, R_DIR, slot) = v.oDir; prec(paths, R_TP, slot) = v.oTp; prec(paths, R_RNG, slot) = v.oRng;
                    nextLive[atomicAdd(&sN[TN_NEXT_LIVE], 1u)] = slot;
                }
                else nextZombies[atomicAdd(&sN[TN_NEXT_ZOMBIES], 1u)] = slot;
                if (ubits(v.oSampler.w) != 0u)
                {
                    prec(paths, R_SH_P, slot) = stage[2][threadIdx.x];
                    pshadow(paths, 0, 0, slot) = stage[0][threadIdx.x];
                    pshadow(paths, 0, 1, slot) = stage[1][threadIdx.x];
                    if (v.rayNeeded) sLists[4][atomicAdd(&sN[TN_SHADOW], 1u)] = slot;
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) { sN[TN_LIVE] = sN[TN_NEXT_LIVE]; sN[TN_ZOMBIES] = sN[TN_NEXT_ZOMBIES]; sN[TN_NEXT_LIVE] = 0u; sN[TN_NEXT_ZOMBIES] = 0u; }
            cur ^= 1u;
            __syncthreads();
        }
        __syncthreads();   // (every thread has left the loop on the same counts before the next chunk resets them)
    }
    flushCounters(cnt, counters);
}

// the instantiations the host side launches
#define RT_X(L, P) template __global__ void RT_TAIL_ATTR(L, P) k_tail<L, P> RT_K_TAIL_ARGS;
RT_K_TAIL_INSTANCES(RT_X)
#undef RT_X
