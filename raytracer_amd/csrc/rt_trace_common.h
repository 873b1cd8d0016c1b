// rt_trace_common.h -- what the traversal kernels (rt_trace.hip), the fused tail kernel (rt_tail.hip) and the host side (rt_runtime.hip) share:
// the work distribution of the persistent kernels and the scheduling knobs the host passes to them.
#pragma once
#include "rt_device_core.h"
#include "rt_device_traverse.h"
#include "rt_device_state.h"

// Work distribution of the persistent traversal kernel.  A wave owns a CHUNK of consecutive queue indices obtained
// with one global atomic and hands them to its idle lanes locally; only when the chunk is used up does it touch
// the global cursor again (a single word sustains only ~88 returning atomics per microsecond).
struct WaveChunk { uint32_t next, end; };

// Gives idle lanes (want == true) indices from the wave's chunk; returns 0xFFFFFFFF for lanes that got none.
RT_DEV uint32_t waveTake(bool want, WaveChunk& chunk)
{
    const unsigned long long mask = __ballot(want);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    const uint32_t available = chunk.end - chunk.next;
    const uint32_t idx = (want && rank < available) ? chunk.next + rank : 0xFFFFFFFFu;
    const uint32_t taken = (uint32_t)__popcll(mask) < available ? (uint32_t)__popcll(mask) : available;
    chunk.next += taken;
    return idx;
}

RT_DEV void waveClaimChunk(WaveChunk& chunk, uint32_t* cursor, uint32_t chunkSize, uint32_t count)
{
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(cursor, chunkSize);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);   // wave-uniform: the chunk lives in scalar registers (every lane is active here; lane 0 claimed)
    chunk.next = base < count ? base : count;
    chunk.end = base + chunkSize < count ? base + chunkSize : count;
    if (chunk.end < chunk.next) chunk.end = chunk.next;
}


#define RT_SPLIT_AFTER 32u   // drain iterations of a wave before its shadow rays start sharing subtrees
#define RT_RETRACE_SPLIT_AFTER 2u   // ... in the re-trace launches behind the 4-wide walks (TravTuning::splitAfter)

// Wave scheduling knobs of the persistent traversal kernel (wave-uniform, passed as kernel arguments)
struct TravTuning
{
    uint32_t refillMinIdle;   // refill once this many lanes of the wave have no ray (or all of them)
    uint32_t otherMinLanes;   // run the "other" phase (leaves, objects, finishing) once this many lanes wait for it
    float shadowOffset;       // any-hit rays start at origin + direction * shadowOffset: 1e-4 (PathTracerMIS.cpp:86, VCM.cpp:673 ...);
                              // 0 for the Light Tracer, whose offset is along the surface normal and already in the stored origin
    uint32_t* overflowQueue;  // closest-hit rays still running this long after the queue ran dry are handed to k_trace_monster
    uint32_t* overflowCount;  // (null: never)
    uint32_t abortClosestAfter;   // ... measured in scheduling rounds of the wave after its queue is exhausted
    const uint32_t* denseCounts;  // dense path state: the closest-hit rays are the live paths of the arena's regions (no queue); else null
    uint32_t denseShardCapacity;
    uint32_t splitAfter;      // drain iterations of a wave before its any-hit rays start sharing subtrees (0: RT_SPLIT_AFTER)
    uint32_t baseBlocks;      // re-trace launches: blocks beyond this many leave at once unless the queues hold more than `fullGridAbove` requests
    uint32_t fullGridAbove;   // (0 / 0: every block works).  A re-trace launch usually holds a few thousand rays -- one block per CU --, but a scene whose sun
                              // shines exactly along an axis hands it a fifth of all next-event rays (tests/test_gpu_parity.py, test_axis_parallel_next_event_rays)
};
#define RT_ABORT_CLOSEST_AFTER 768u
// the same hand-over in the re-trace launches behind the 4-wide walks (PathTracerMIS) and in a block's local second walk: their queues hold a few
// thousand rays, a wave's queue is dry after its first claim, an ordinary ray is done within ~10 rounds (a round = a run of interior steps + a leaf)
#define RT_ABORT_RETRACE_AFTER 96u

#define RT_COUNTER_RETRACED 12   // counters[]: rays the 4-wide walks handed to the binary-tree walk (RtCounters::numRetracedRays)
#define RT_MONSTER_BLOCK 512
#define RT_MONSTER_STACK 8192
