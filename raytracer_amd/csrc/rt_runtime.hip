// rt_runtime.hip -- the host side of the C-ABI of include/rtgpu.h: contexts, batch lanes, path-state arenas, scene upload, the launch
// sequences of the integrators.  The kernels live in rt_trace.hip (traversal, camera rays, post-process, known-answer hooks), rt_shade.hip
// (shading) and rt_tail.hip (the fused tail of a batch); this unit launches them through rt_trace_kernels.h / rt_shade_kernels.h.
//
// One pass (= one sample per owned pixel) is a fixed sequence of launches on the context's stream:
//
//   generate                               camera ray + sampler reset per path            (Viewport.cpp:305-331)
//   for depth = 0 .. maxRayDepth:
//       trace_closest                      two-level BVH closest hit                      (Scene.cpp:219)
//       shade                              miss / light hit / emission / NEE set-up / Russian roulette /
//                                          BSDF sample; compacts survivors into the next queue
//                                                                                         (PathTracerMIS.cpp:270-396)
//       trace_shadow                       any-hit occlusion of the NEE rays + accumulate (PathTracerMIS.cpp:81-119)
//   accumulate                             film sum (+ secondary sum on even passes)      (Film.cpp:25-39)
//
// Path state lives in HBM as structure-of-arrays indexed by path slot, so a wave reads 64 consecutive
// dwords per field; queues hold slot indices and are compacted with wave ballots (one atomic per wave).
// The two traversal kernels are PERSISTENT: a fixed grid of waves pulls rays from the queue through an atomic
// cursor and refills lanes whose ray has finished (rays of one wave take very different numbers of node
// steps), with the per-lane node stack in LDS.  NEE rays go through their own dense queue; their results are
// folded into the path radiance by the next kernel that touches the path (shade of the next bounce, or
// accumulate), which preserves the reference's accumulation order.
// Per-path arithmetic is kept in the reference's operation order (see rt_device_math.h), which makes the
// result independent of the wavefront schedule and reproducible against the CPU oracle.
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include "rt_trace_common.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <string>
#include <unordered_map>
#include <vector>

using namespace rtd;

#define RT_HOST_BUILDERS 1
#include "rt_wide_grid.inl"
#include "rt_trace_wide.inl"
#include "rt_trace_wide2.inl"
#include "rt_trace_kernels.h"
#include "rt_shade_kernels.h"
#include "rt_tail_kernels.h"

// =====================================================================================================
// Host side of the C-ABI
// =====================================================================================================
// Synchronous copies between HOST memory the caller owns and the device.  HIP would page-lock a large pageable range on the fly and keep the
// registration cached; the caller then frees the range (a std::vector of the host mirror, a numpy array) and a later allocation lands on the
// same addresses -- on some boxes of the pool the HSA runtime aborts the process a few dozen contexts later (no message; it went away with
// this).  So anything above 64 KB that is not page-locked already (hipHostMalloc / hipHostRegister: the viewport's sum bitmaps) travels
// through a page-locked staging buffer of the library, 8 MB at a time.
#include <mutex>
// one staging buffer (and its lock) per device: contexts on different devices copy side by side (rtgpu_create_multi).  8 MB of page-locked memory
// per device used, kept for the life of the process (freeing it from an exit handler would race the HIP runtime's own teardown)
struct Staging { std::mutex mutex; void* buffer = nullptr; };
static std::mutex gStagingTableMutex;
static std::unordered_map<int, Staging*> gStaging;
static const size_t kStagingBytes = (size_t)8 << 20;
static Staging* stagingOfCurrentDevice()
{
    int device = 0;
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> lock(gStagingTableMutex);
    Staging*& s = gStaging[device];
    if (!s) s = new Staging();
    return s;
}
static bool hostRangeIsPageLocked(const void* p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // ordinary pageable memory: "invalid value"
    return attr.type == hipMemoryTypeHost;
}
static hipError_t rtMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
    if (bytes == 0) return hipSuccess;
    const bool h2d = kind == hipMemcpyHostToDevice, d2h = kind == hipMemcpyDeviceToHost;
    if ((!h2d && !d2h) || bytes <= ((size_t)64 << 10) || hostRangeIsPageLocked(h2d ? src : dst)) return hipMemcpy(dst, src, bytes, kind);
    Staging* const st = stagingOfCurrentDevice();
    std::lock_guard<std::mutex> lock(st->mutex);
    if (!st->buffer)
    {
        const hipError_t e = hipHostMalloc(&st->buffer, kStagingBytes, hipHostMallocPortable);
        if (e != hipSuccess) { st->buffer = nullptr; return e; }
    }
    for (size_t done = 0; done < bytes; done += kStagingBytes)
    {
        const size_t n = bytes - done < kStagingBytes ? bytes - done : kStagingBytes;
        if (h2d) memcpy(st->buffer, static_cast<const char*>(src) + done, n);
        const hipError_t e = h2d ? hipMemcpy(static_cast<char*>(dst) + done, st->buffer, n, kind) : hipMemcpy(st->buffer, static_cast<const char*>(src) + done, n, kind);
        if (e != hipSuccess) return e;
        if (d2h) memcpy(static_cast<char*>(dst) + done, st->buffer, n);
    }
    return hipSuccess;
}

static thread_local std::string gLastError;

static int fail(int code, const std::string& msg) { gLastError = msg; return code; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(_e == hipErrorOutOfMemory ? RTGPU_ERR_OUT_OF_MEMORY : RTGPU_ERR_DEVICE,          \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                             \
    } while (0)

enum KernelClass { KC_GENERATE = 0, KC_TRACE, KC_SHADE, KC_ACCUMULATE, KC_RETRACE, KC_TAIL, KC_COUNT };
static const char* const kKernelClassNames[RTGPU_NUM_KERNEL_CLASSES] = { "generate", "trace", "shade", "accumulate", "retrace", "tail", "", "" };

#define RT_SEED_RING 128

struct CtxPending { DevPass pass; std::vector<uint32_t> seeds; };
#define RT_VCM_MAX_BATCH 8

#define RT_MAX_LANES 6
struct BatchLane
{
    hipStream_t stream = nullptr;
    Paths paths = { nullptr, 0, 0 };
    uint32_t* queues[2] = { nullptr, nullptr };
    uint32_t* shadowQueues[2] = { nullptr, nullptr };   // capacity * maxLights NEE ray requests each, ping-pong per bounce
    uint32_t* exactQueue = nullptr;        // closest-hit rays / any-hit requests the 4-wide walks hand to the binary-tree kernel
    uint32_t* exactShadowQueue = nullptr;
    // dense path state (rt_dense.inl, LightSamplingStrategy::Single): the second arena of the ping-pong, the parked radiance of
    // finished paths, per bounce the live / zombie counts of the arena's regions (2 * RT_DENSE_SHARDS words per bounce)
    Paths paths2 = { nullptr, 0, 0 };
    float4* home = nullptr; size_t homeCapacity = 0;
    uint32_t* denseCounts = nullptr;
    // per-batch work counters, 8 planes of (maxDepth + 2) uint32, zeroed once per batch: path-queue counts,
    // shadow-queue counts, traversal cursors, -, exact-queue counts, exact-shadow-queue counts, exact cursors, - (one of each per
    // bounce, so that no reset ever races with a reader)
    uint32_t* queueCounts = nullptr;
    uint32_t queueCountCapacity = 0;
    hipEvent_t accumulated = nullptr;   // recorded after the lane's k_accumulate
};

struct RtgpuContext
{
    int device = 0;
    uint32_t numCUs = 256;

    // scene (device copies); sceneDev holds DEVICE pointers
    RtSceneDesc sceneDev;
    std::vector<void*> sceneAllocs;
    bool sceneReady = false;
    uint32_t numLights = 0;

    // film
    uint32_t width = 0, height = 0;
    RtgpuShard shard = { 0, 1 };
    // rtgpu_create_multi: the context the caller holds renders shard 0 and owns one more context per further device (shards 1..);
    // every call fans out, the read-back calls gather the peers' tiles into this context's sum buffers first (rt_multi.inl)
    std::vector<RtgpuContext*> peers;
    bool isPeer = false;
    bool stagedGather = false;         // no peer access between the devices (or RTGPU_MULTI_STAGED=1): hipMemcpyPeerAsync into staging buffers, then the gather
    float* gatherStage = nullptr; size_t gatherStageFloats = 0;
    bool axisParallelSun = false;      // the scene has a delta directional light along a coordinate plane / axis: its next-event rays fill the re-trace launches (full grid there)
    RtMultiInfo multiInfo = {};        // rtgpu_get_multi_info: which gather was chosen and why, its timings
    float* sum = nullptr;
    float* secondary = nullptr;
    uint32_t* slotPixel = nullptr;
    uint32_t numSlots = 0;
    std::vector<uint8_t> activeMask;   // adaptive rendering: 1 = pixel inside an active block; empty = whole image

    // Batch lanes.  Every batch of passes runs on ONE lane = its own stream, path-state arena, queues and work
    // counters; consecutive batches alternate lanes, so the drain of a persistent traversal launch (a handful of
    // rays with thousands of steps keep a few waves busy for milliseconds -- an axis-parallel NEE ray that grazes
    // a plane of box faces can take 30 000) overlaps with the next batch's kernels instead of idling the chip.
    // Only k_accumulate is ordered across lanes (an event): the film is summed in pass order.
    BatchLane lanes[RT_MAX_LANES];
    uint32_t numLanes = 4;
    bool lanesChosen = false;          // by RTGPU_LANES or rtgpu_set_concurrency; otherwise shards (< 1.1 M owned pixels) run 4 lanes
    uint32_t nextLane = 0;
    int lastAccumulateLane = -1;
    uint32_t traversalStackNeed = 0;   // deepest top-level + mesh stack the uploaded scene can produce
    WideBvh wide;                      // 4-wide collapse of the same tree (rt_trace_wide.inl); nodes == nullptr: none
    WideScene wide2;                   // two-level scenes: 4-wide top-level tree over 4-wide mesh trees (rt_trace_wide2.inl); nodes == nullptr: none
    bool wide2Allowed = true;          // RTGPU_WIDE2=0: two-level scenes keep the binary walk
    uint64_t walkNodeBytes[3] = { 0, 0, 0 }, walkLeafBoxBytes[3] = { 0, 0, 0 }, walkTriangleBytes = 0;   // rtgpu_get_walk_info, per RTGPU_WALK_* kernel
    bool wideAllowed = true;           // RTGPU_WIDE=0: single-mesh scenes walk the binary tree (k_trace) even with the intersection counters off
    bool denseAllowed = true;          // RTGPU_NO_DENSE=1: path state stays in the pixel's slot for the whole path (the first layout)
    TravTuning tune = { 28u, 32u, 0.0001f, nullptr, nullptr, RT_ABORT_CLOSEST_AFTER, nullptr, 0u };   // scheduling: measured plateau on MI355X (profiles/r01_tuning_sweep.txt)
    uint32_t travBlocksPerCU = 0;      // 0 = default
    int32_t tailBounce = -1;           // rtgpu_set_schedule: the bounce at which a dense batch hands over to k_tail (rt_tail.hip); 0 = never, -1 = policy
    int32_t localRetrace = -1;         // rtgpu_set_schedule: the 4-wide walks trace their undecided rays themselves; 0 / 1, -1 = policy
    int leanScene = 0;                 // the scene class of rt_device_core.h (kLean): 0 anything, 1 lean, 2 lean + textures, 3 anything without textures, 4 lean + simple bitmaps only
    bool countIntersections = false;   // box / triangle test counters: RT_ENABLE_INTERSECTION_COUNTERS of the reference, off by default like there (Core/Config.h:4);
                                       // rtgpu_set_intersection_counters, or RTGPU_INTERSECTION_COUNTERS=1 for the default of new contexts
    unsigned long long* counters = nullptr;   // 16 x u64
    uint32_t* deviceFlags = nullptr;          // page-locked, device-visible: kernels raise [0] when a region of a dense arena overflows; checked by every synchronising call

    // passes queued by rtgpu_render_pass and not yet submitted: up to passBatch of them ride through ONE launch
    // sequence (their paths are simply more slots), which amortises the per-launch tail of the persistent kernels
    std::vector<CtxPending> pending;
    uint32_t passBatch = 8;
    bool passBatchFromEnv = false;     // otherwise small frames / small shards (< 400 k owned pixels) batch 16 passes
    // A caller that streams passes (no read-back in between) gets larger batches: after every submitted batch of a full-size frame
    // the next one grows by 8 passes up to 24 (8 -> 2100, 16 -> 2125-2190, 24 -> 2195-2210 Msamples/s over 256 passes); any
    // synchronising call starts over at the base size, so a caller that renders few passes between read-backs keeps the small batches.
    uint32_t passBatchBase = 8;
    size_t laneBudgetBytes = (size_t)32 << 30;   // device memory one batch lane may take: 32 GB, less on a device that could not hold four such lanes
    uint32_t batchesAtThisSize = 0;    // full batches submitted at the current passBatch
    uint32_t batchesSinceSync = 0;     // batches submitted since the last synchronising call (their lanes are busy)
    DevPass* passRingDev = nullptr;
    DevPass* passRingHost = nullptr;    // pinned

    // per-pass seed ring
    uint32_t* seedRingDev = nullptr;
    uint32_t* seedRingHost = nullptr;   // pinned
    hipEvent_t seedEvents[RT_SEED_RING];
    bool seedEventUsed[RT_SEED_RING];
    uint32_t seedCursor = 0;

    bool plainPathTracer = false;      // RT_INTEGRATOR_PATH_TRACER: k_shade<false, true>
    bool lightTracer = false;          // RT_INTEGRATOR_LIGHT_TRACER: the light stage of rt_vcm.inl without MIS (k_lt_shade)
    int debugMode = -1;                // RT_INTEGRATOR_DEBUG: DebugRenderingMode, k_debug_shade after the primary rays' traversal
    // bidirectional integrator (rt_vcm.inl) on lane 0's stream.  Like PathTracerMIS passes, VCM passes ride through the launch
    // sequence in batches: the light stages of the batch first (pass j's photons are the merge set of pass j+1, so the hash grids
    // of passes 1.. are built between the stages), then the camera stages -- the same results as one pass at a time.
    struct Vcm
    {
        bool enabled = false;
        RtVcmParams params;
        float mergingRadiusVC = 0.0f, mergingRadiusVM = 0.0f;
        Paths lightPaths = { nullptr, 0, 0 }, cameraPaths = { nullptr, 0, 0 };
        VcmArena arena = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0 };
        uint32_t* mergeQueue = nullptr; uint32_t* connectQueue = nullptr;
        uint32_t* overflowQueue = nullptr;   // closest-hit rays k_trace hands to k_trace_monster
        uint32_t* exactQueue = nullptr; uint32_t* exactShadowQueue = nullptr;   // what the 4-wide walks hand to the binary-tree kernel (as BatchLane's)
        uint32_t traceSerial = 0;            // trace launches since the counters were zeroed: every launch has its own hand-over counters
        uint32_t* queues[4] = { nullptr, nullptr, nullptr, nullptr };          // light ping-pong, camera ping-pong
        uint32_t* shadowQueues[4] = { nullptr, nullptr, nullptr, nullptr };
        uint32_t* counts = nullptr;                                               // 6 planes of RT_VCM_COUNT_PLANE
        DevPass* passDev = nullptr; uint32_t* seedDev = nullptr;                 // RT_VCM_MAX_BATCH entries each
        VcmDev* devsDev = nullptr; HashGridView* gridsDev = nullptr;
        VcmPhotonGrid grids[RT_VCM_MAX_BATCH];                                    // merge set of pass j of the batch
        bool havePhotons = false;     // the arena holds the photons of the pass before the next one ...
        uint32_t lastPhotonPass = 0;  // ... in the storage of this pass of the last batch
        uint32_t requestsPerVertex = 0;
        uint32_t batch = 1, batchCapacity = 0;   // passes per launch sequence; what the arenas were sized for
        // passes queued by rtgpu_render_pass and not yet submitted
        struct Pending { RtPassParams params; std::vector<uint32_t> seeds; };
        std::vector<Pending> pending;
    } vcm;

    // timing
    bool timing = false;
    struct Timed { int kc; hipEvent_t a, b; };
    std::vector<Timed> pendingTimed;
    std::vector<hipEvent_t> eventPool;
    double kernelMs[RTGPU_NUM_KERNEL_CLASSES];
    uint64_t kernelLaunches[RTGPU_NUM_KERNEL_CLASSES];
};

static void freeScene(RtgpuContext* c)
{
    for (void* p : c->sceneAllocs) (void)hipFree(p);
    c->sceneAllocs.clear();
    memset(&c->sceneDev, 0, sizeof(c->sceneDev));
    c->sceneReady = false;
}

static void freeFilm(RtgpuContext* c)
{
    if (c->sum) (void)hipFree(c->sum);
    if (c->secondary) (void)hipFree(c->secondary);
    if (c->slotPixel) (void)hipFree(c->slotPixel);
    c->sum = c->secondary = nullptr; c->slotPixel = nullptr; c->numSlots = 0;
}

static void freePaths(BatchLane& l)
{
    if (l.paths.base) (void)hipFree(l.paths.base);
    if (l.queues[0]) (void)hipFree(l.queues[0]);
    if (l.queues[1]) (void)hipFree(l.queues[1]);
    if (l.shadowQueues[0]) (void)hipFree(l.shadowQueues[0]);
    if (l.shadowQueues[1]) (void)hipFree(l.shadowQueues[1]);
    if (l.exactQueue) (void)hipFree(l.exactQueue);
    if (l.exactShadowQueue) (void)hipFree(l.exactShadowQueue);
    if (l.paths2.base) (void)hipFree(l.paths2.base);
    if (l.home) (void)hipFree(l.home);
    l.exactQueue = l.exactShadowQueue = nullptr; l.paths2.base = nullptr; l.paths2.capacity = 0; l.paths2.maxLights = 0; l.home = nullptr; l.homeCapacity = 0;
    l.paths.base = nullptr; l.paths.capacity = 0; l.paths.maxLights = 0;
    l.queues[0] = l.queues[1] = nullptr; l.shadowQueues[0] = l.shadowQueues[1] = nullptr;
}

static hipError_t syncLanes(RtgpuContext* c)
{
    hipError_t first = hipSuccess;
    for (uint32_t i = 0; i < RT_MAX_LANES; ++i)
        if (c->lanes[i].stream) { const hipError_t e = hipStreamSynchronize(c->lanes[i].stream); if (first == hipSuccess) first = e; }
    return first;
}

static int resolveTimed(RtgpuContext* c)
{
    for (auto& t : c->pendingTimed)
    {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
        c->kernelMs[t.kc] += ms;
        c->kernelLaunches[t.kc]++;
        c->eventPool.push_back(t.a); c->eventPool.push_back(t.b);
    }
    c->pendingTimed.clear();
    return RTGPU_OK;
}

static hipEvent_t acquireEvent(RtgpuContext* c)
{
    if (!c->eventPool.empty()) { hipEvent_t e = c->eventPool.back(); c->eventPool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct LaunchTimer
{
    RtgpuContext* c; hipStream_t stream; int kc; hipEvent_t a = nullptr, b = nullptr;
    LaunchTimer(RtgpuContext* ctx, hipStream_t st, int k) : c(ctx), stream(st), kc(k)
    {
        if (c->timing) { a = acquireEvent(c); b = acquireEvent(c); (void)hipEventRecord(a, stream); }
    }
    ~LaunchTimer()
    {
        if (c->timing) { (void)hipEventRecord(b, stream); c->pendingTimed.push_back({ kc, a, b }); }
    }
};

static int flushPending(RtgpuContext* c);
static void freeVcm(RtgpuContext* c);

template <typename T>
static int uploadArray(RtgpuContext* c, const T* host, size_t count, const T** outDev)
{
    *outDev = nullptr;
    if (count == 0) return RTGPU_OK;
    if (!host) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene array pointer is NULL but its count is not zero");
    void* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, count * sizeof(T)));
    c->sceneAllocs.push_back(dev);
    HIP_TRY(rtMemcpy(dev, host, count * sizeof(T), hipMemcpyHostToDevice));
    *outDev = static_cast<const T*>(dev);
    return RTGPU_OK;
}

// depth of a BVH in stack entries: the traversal pushes at most one node per interior level
static uint32_t bvhDepth(const RtNode* nodes, uint32_t numNodes)
{
    if (numNodes == 0) return 0;
    uint32_t maxDepth = 0;
    std::vector<std::pair<uint32_t, uint32_t>> stack;
    stack.push_back({ 0u, 0u });
    while (!stack.empty())
    {
        const auto [idx, depth] = stack.back(); stack.pop_back();
        if (idx >= numNodes) return 0xFFFFFFFFu;
        const RtNode& n = nodes[idx];
        if ((n.leaves & 0x3FFFFFFFu) != 0) { if (depth > maxDepth) maxDepth = depth; continue; }
        if (depth > 4096) return 0xFFFFFFFFu;
        stack.push_back({ n.childIndex, depth + 1 }); stack.push_back({ n.childIndex + 1, depth + 1 });
    }
    return maxDepth;
}

// round trip of a host buffer through one of the KAT kernels (synchronous, lane 0's stream)
template <typename Launch>
static int katRoundTrip(RtgpuContext* c, const void* in, size_t inBytes, void* out, size_t outBytes, Launch launch)
{
    HIP_TRY(hipSetDevice(c->device));
    void* dIn = nullptr; void* dOut = nullptr;
    hipError_t e = hipMalloc(&dIn, inBytes ? inBytes : 4);
    if (e == hipSuccess) e = hipMalloc(&dOut, outBytes ? outBytes : 4);
    if (e == hipSuccess) e = rtMemcpy(dIn, in, inBytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(dOut, 0, outBytes, c->lanes[0].stream);   // on the kernel's stream: the lanes do not synchronise with the null stream
    if (e == hipSuccess)
    {
        launch(dIn, dOut, c->lanes[0].stream);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->lanes[0].stream);
    }
    if (e == hipSuccess) e = rtMemcpy(out, dOut, outBytes, hipMemcpyDeviceToHost);
    if (dIn) (void)hipFree(dIn);
    if (dOut) (void)hipFree(dOut);
    if (e != hipSuccess) return fail(RTGPU_ERR_DEVICE, std::string("rtgpu_kat: ") + hipGetErrorString(e));
    return RTGPU_OK;
}

#include "rt_multi.inl"

// Streams are recycled through a process-wide pool instead of being created and destroyed with every context: a test session (or an
// application that opens a renderer per frame size) goes through hundreds of contexts, and on some boxes of the pool the HSA runtime's
// event thread aborts the process after a few hundred stream (hardware queue) create / destroy cycles (no message; ROCm 7.0.2).  A context
// returns its idle streams at destruction, after it has synchronised them.
#include <mutex>
static std::mutex gStreamPoolMutex;
static std::unordered_map<int, std::vector<hipStream_t>> gStreamPool;   // device -> idle non-blocking streams
static hipError_t acquireStream(int device, hipStream_t* out)
{
    {
        std::lock_guard<std::mutex> lock(gStreamPoolMutex);
        auto& pool = gStreamPool[device];
        if (!pool.empty()) { *out = pool.back(); pool.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
static void releaseStream(int device, hipStream_t stream)
{
    std::lock_guard<std::mutex> lock(gStreamPoolMutex);
    gStreamPool[device].push_back(stream);
}

extern "C" {

#define RTGPU_API __attribute__((visibility("default")))

RTGPU_API const char* rtgpu_last_error(void) { return gLastError.c_str(); }
RTGPU_API uint32_t rtgpu_abi_version(void) { return RTGPU_ABI_VERSION; }

RTGPU_API int rtgpu_create(int deviceIndex, RtgpuContext** outCtx)
{
    if (!outCtx) return fail(RTGPU_ERR_INVALID_ARGUMENT, "outCtx is NULL");
    *outCtx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RTGPU_ERR_NO_DEVICE, "no HIP device available");
    if (deviceIndex < 0 || deviceIndex >= n) return fail(RTGPU_ERR_INVALID_ARGUMENT, "device index out of range");
    HIP_TRY(hipSetDevice(deviceIndex));
    RtgpuContext* c = new RtgpuContext();
    c->device = deviceIndex;
    memset(&c->sceneDev, 0, sizeof(c->sceneDev));
    memset(c->kernelMs, 0, sizeof(c->kernelMs)); memset(c->kernelLaunches, 0, sizeof(c->kernelLaunches));
    for (int i = 0; i < RT_SEED_RING; ++i) { c->seedEvents[i] = nullptr; c->seedEventUsed[i] = false; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, deviceIndex) == hipSuccess) c->numCUs = (uint32_t)prop.multiProcessorCount;
    {
        // six lanes (the most rtgpu_set_concurrency allows) plus scene, film and the bidirectional integrator's arenas must fit what is free now
        size_t freeBytes = 0, totalBytes = 0;
        if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess && freeBytes / 8u < c->laneBudgetBytes) c->laneBudgetBytes = freeBytes / 8u > ((size_t)64 << 20) ? freeBytes / 8u : ((size_t)64 << 20);
        // RTGPU_LANE_BUDGET_MB: the device memory ONE batch lane may take for its path-state arenas (a co-tenant's knob: four lanes of a full-HD frame reach
        // ~70 GB with the default; 4096 holds them to 16 GB at 5-pass batches).  Only ever lowers the budget; results do not depend on it (the batch a lane
        // holds shrinks, maxBatchFor / ensurePaths).
        if (const char* e = getenv("RTGPU_LANE_BUDGET_MB"))
        {
            const size_t asked = (size_t)strtoull(e, nullptr, 10) << 20;
            if (asked >= ((size_t)64 << 20) && asked < c->laneBudgetBytes) c->laneBudgetBytes = asked;
        }
    }
    // scheduling knobs (performance only; results do not depend on them)
    if (const char* e = getenv("RTGPU_REFILL_MIN_IDLE")) c->tune.refillMinIdle = (uint32_t)atoi(e);
    if (const char* e = getenv("RTGPU_OTHER_MIN_LANES")) c->tune.otherMinLanes = (uint32_t)atoi(e);
    if (const char* e = getenv("RTGPU_TRAV_BLOCKS_PER_CU")) c->travBlocksPerCU = (uint32_t)atoi(e);
    if (const char* e = getenv("RTGPU_WIDE")) c->wideAllowed = atoi(e) != 0;
    if (const char* e = getenv("RTGPU_INTERSECTION_COUNTERS")) c->countIntersections = atoi(e) != 0;
    memset(&c->wide, 0, sizeof(c->wide));
    if (const char* e = getenv("RTGPU_NO_DENSE")) c->denseAllowed = atoi(e) == 0;
    if (const char* e = getenv("RTGPU_WIDE2")) c->wide2Allowed = atoi(e) != 0;
    if (const char* e = getenv("RTGPU_PASS_BATCH")) { c->passBatch = (uint32_t)atoi(e); c->passBatchFromEnv = true; }
    if (c->passBatch < 1) c->passBatch = 1;
    if (c->passBatch > RT_SEED_RING / 2) c->passBatch = RT_SEED_RING / 2;
    if (const char* e = getenv("RTGPU_LANES")) { c->numLanes = (uint32_t)atoi(e); c->lanesChosen = true; }
    if (c->numLanes < 1) c->numLanes = 1;
    if (c->numLanes > RT_MAX_LANES) c->numLanes = RT_MAX_LANES;
    if (c->tune.refillMinIdle < 1) c->tune.refillMinIdle = 1;
    if (c->tune.otherMinLanes < 1) c->tune.otherMinLanes = 1;
    hipError_t e = hipSuccess;
    for (uint32_t i = 0; i < RT_MAX_LANES && e == hipSuccess; ++i)
    {
        e = acquireStream(c->device, &c->lanes[i].stream);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->lanes[i].accumulated, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&c->counters, 16 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(c->counters, 0, 16 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->deviceFlags, 16 * sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess) memset(c->deviceFlags, 0, 16 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e == hipSuccess) e = hipMalloc((void**)&c->seedRingDev, (size_t)RT_SEED_RING * RTGPU_MAX_DIMENSIONS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->seedRingHost, (size_t)RT_SEED_RING * RTGPU_MAX_DIMENSIONS * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&c->passRingDev, (size_t)RT_SEED_RING * sizeof(DevPass));
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->passRingHost, (size_t)RT_SEED_RING * sizeof(DevPass), hipHostMallocDefault);
    for (int i = 0; i < RT_SEED_RING && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->seedEvents[i], hipEventDisableTiming);
    if (e != hipSuccess)
    {
        const std::string msg = std::string("context creation failed: ") + hipGetErrorString(e);
        rtgpu_destroy(c);
        return fail(RTGPU_ERR_DEVICE, msg);
    }
    *outCtx = c;
    return RTGPU_OK;
}

RTGPU_API int rtgpu_create_multi(const int* deviceIndices, uint32_t numDevices, RtgpuContext** outCtx)
{
    if (!outCtx) return fail(RTGPU_ERR_INVALID_ARGUMENT, "outCtx is NULL");
    *outCtx = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return fail(RTGPU_ERR_NO_DEVICE, "no HIP device available");
    std::vector<int> devices;
    if (deviceIndices) devices.assign(deviceIndices, deviceIndices + numDevices);
    else for (int d = 0; d < (numDevices ? (int)numDevices : visible); ++d) devices.push_back(d);   // NULL: the first numDevices (0: all) visible ones
    if (devices.empty() || devices.size() > RTGPU_MAX_DEVICES) return fail(RTGPU_ERR_INVALID_ARGUMENT, "1..16 devices");
    for (int d : devices) if (d < 0 || d >= visible) return fail(RTGPU_ERR_INVALID_ARGUMENT, "device index out of range");
    RtgpuContext* c = nullptr;
    int r = rtgpu_create(devices[0], &c); if (r) return r;
    const uint32_t world = (uint32_t)devices.size();
    c->shard = { 0u, world };
    if (const char* e = getenv("RTGPU_MULTI_STAGED")) c->stagedGather = atoi(e) != 0;
    RtMultiInfo& info = c->multiInfo;
    info.numDevices = world; info.reasonDevice = -1; info.gatherReason = c->stagedGather ? 1u : 0u;
    for (uint32_t k = 0; k < world; ++k) { info.devices[k] = devices[k]; info.peerAccess[k] = devices[k] == devices[0] ? 1u : 0u; }
    for (uint32_t k = 1; k < world; ++k)
    {
        RtgpuContext* p = nullptr;
        r = rtgpu_create(devices[k], &p);
        if (r) { rtgpu_destroy(c); return r; }
        p->shard = { k, world }; p->isPeer = true;
        c->peers.push_back(p);
        if (devices[k] != devices[0] && !c->stagedGather)
        {
            // the gather kernel on device 0 reads the peers' sum buffers in place
            int can = 0;
            (void)hipSetDevice(devices[0]);
            if (hipDeviceCanAccessPeer(&can, devices[0], devices[k]) != hipSuccess || !can) { c->stagedGather = true; info.gatherReason = 2u; info.reasonDevice = devices[k]; }
            else
            {
                const hipError_t e = hipDeviceEnablePeerAccess(devices[k], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { c->stagedGather = true; info.gatherReason = 3u; info.reasonDevice = devices[k]; info.reasonError = (int32_t)e; }
                else info.peerAccess[k] = 1u;
                (void)hipGetLastError();
            }
        }
    }
    (void)hipSetDevice(devices[0]);
    info.gatherMode = world == 1u ? RTGPU_GATHER_NONE : (c->stagedGather ? RTGPU_GATHER_STAGED_COPY : RTGPU_GATHER_PEER_KERNEL);
    if (getenv("RTGPU_VERBOSE") && atoi(getenv("RTGPU_VERBOSE")) != 0)
    {
        static const char* const why[] = { "", " (RTGPU_MULTI_STAGED=1)", " (hipDeviceCanAccessPeer: no)", " (hipDeviceEnablePeerAccess failed)" };
        fprintf(stderr, "[rtgpu] multi-device context: %u devices [", world);
        for (uint32_t k = 0; k < world; ++k) fprintf(stderr, "%s%d", k ? " " : "", devices[k]);
        fprintf(stderr, "], read-back gather = %s%s\n", world == 1u ? "none" : (c->stagedGather ? "hipMemcpyPeerAsync into staging buffers + kernel" : "kernel reading the peers' buffers in place (peer access)"),
                why[info.gatherReason < 4u ? info.gatherReason : 0u]);
    }
    *outCtx = c;
    return RTGPU_OK;
}

RTGPU_API int rtgpu_get_multi_info(RtgpuContext* c, RtMultiInfo* out)
{
    if (!c || !out) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    *out = c->multiInfo;
    if (out->numDevices == 0u) { out->numDevices = 1u; out->devices[0] = c->device; out->peerAccess[0] = 1u; out->reasonDevice = -1; }   // a one-device context (rtgpu_create)
    return RTGPU_OK;
}

RTGPU_API int rtgpu_num_devices(RtgpuContext* c, uint32_t* outCount)
{
    if (!c || !outCount) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    *outCount = (uint32_t)c->peers.size() + 1u;
    return RTGPU_OK;
}

RTGPU_API void rtgpu_destroy(RtgpuContext* c)
{
    if (!c) return;
    for (RtgpuContext* p : c->peers) rtgpu_destroy(p);
    c->peers.clear();
    (void)hipSetDevice(c->device);
    (void)syncLanes(c);
    if (c->gatherStage) (void)hipFree(c->gatherStage);
    freeScene(c); freeFilm(c);
    for (uint32_t i = 0; i < RT_MAX_LANES; ++i)
    {
        freePaths(c->lanes[i]);
        if (i == 0) freeVcm(c);
        if (c->lanes[i].queueCounts) (void)hipFree(c->lanes[i].queueCounts);
        if (c->lanes[i].denseCounts) (void)hipFree(c->lanes[i].denseCounts);
        if (c->lanes[i].accumulated) (void)hipEventDestroy(c->lanes[i].accumulated);
    }
    if (c->counters) (void)hipFree(c->counters);
    if (c->deviceFlags) (void)hipHostFree(c->deviceFlags);
    if (c->seedRingDev) (void)hipFree(c->seedRingDev);
    if (c->seedRingHost) (void)hipHostFree(c->seedRingHost);
    if (c->passRingDev) (void)hipFree(c->passRingDev);
    if (c->passRingHost) (void)hipHostFree(c->passRingHost);
    for (int i = 0; i < RT_SEED_RING; ++i) if (c->seedEvents[i]) (void)hipEventDestroy(c->seedEvents[i]);
    for (auto& t : c->pendingTimed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (hipEvent_t e : c->eventPool) (void)hipEventDestroy(e);
    for (uint32_t i = 0; i < RT_MAX_LANES; ++i) if (c->lanes[i].stream) { (void)hipStreamSynchronize(c->lanes[i].stream); releaseStream(c->device, c->lanes[i].stream); }
    delete c;
}

RTGPU_API int rtgpu_upload_scene(RtgpuContext* c, const RtSceneDesc* s)
{
    if (!c || !s) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    RT_FAN_OUT(c, rtgpu_upload_scene(peer, s));   // the scene is replicated: every device traverses its own copy
    if (s->abiVersion != RTGPU_ABI_VERSION) return fail(RTGPU_ERR_INVALID_ARGUMENT, "RtSceneDesc::abiVersion mismatch");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = flushPending(c); if (fr) return fr; }
    HIP_TRY(syncLanes(c));

    // validation: indices in range, stacks deep enough
    if (s->numObjects > 1 && s->numTopNodes == 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "scene with more than one object needs a top-level BVH");
    const uint32_t topDepth = bvhDepth(s->topNodes, s->numTopNodes);
    uint32_t maxMeshDepth = 0;
    if (topDepth == 0xFFFFFFFFu) return fail(RTGPU_ERR_INVALID_ARGUMENT, "malformed top-level BVH");
    for (uint32_t i = 0; i < s->numMeshes; ++i)
    {
        const RtMesh& m = s->meshes[i];
        if ((uint64_t)m.firstNode + m.numNodes > s->numMeshNodes || (uint64_t)m.firstTriangle + m.numTriangles > s->numTriangles || (uint64_t)m.firstVertex + m.numVertices > s->numVertices)
            return fail(RTGPU_ERR_INVALID_ARGUMENT, "mesh ranges out of bounds");
        for (uint32_t t = 0; t < m.numTriangles; ++t)
        {
            const RtVertexIndices& idx = s->vertexIndices[m.firstTriangle + t];
            if (idx.i0 >= m.numVertices || idx.i1 >= m.numVertices || idx.i2 >= m.numVertices) return fail(RTGPU_ERR_INVALID_ARGUMENT, "triangle vertex index out of range");
        }
        const uint32_t md = bvhDepth(s->meshNodes + m.firstNode, m.numNodes);
        if (md == 0xFFFFFFFFu) return fail(RTGPU_ERR_INVALID_ARGUMENT, "malformed mesh BVH");
        if (md > maxMeshDepth) maxMeshDepth = md;
    }
    if (topDepth + maxMeshDepth > 64) return fail(RTGPU_ERR_UNSUPPORTED, "BVH deeper than the 64-entry traversal stack");
    for (uint32_t i = 0; i < s->numTopNodes; ++i) if ((s->topNodes[i].leaves & 0x3FFFFFFFu) > RT_MAX_PACKED_LEAVES) return fail(RTGPU_ERR_UNSUPPORTED, "BVH leaves with more than 3 items are not supported");
    for (uint32_t i = 0; i < s->numMeshNodes; ++i) if ((s->meshNodes[i].leaves & 0x3FFFFFFFu) > RT_MAX_PACKED_LEAVES) return fail(RTGPU_ERR_UNSUPPORTED, "BVH leaves with more than 3 items are not supported");
    for (uint32_t i = 0; i < s->numObjects; ++i)
    {
        const RtObject& o = s->objects[i];
        if (o.objectKind == RT_OBJECT_LIGHT) { if (o.lightIndex >= s->numLights) return fail(RTGPU_ERR_INVALID_ARGUMENT, "object light index out of range"); }
        else
        {
            if (o.materialIndex >= s->numMaterials) return fail(RTGPU_ERR_INVALID_ARGUMENT, "object material index out of range");
            if (o.shapeKind == RT_SHAPE_MESH && o.meshIndex >= s->numMeshes) return fail(RTGPU_ERR_INVALID_ARGUMENT, "object mesh index out of range");
            if (o.shapeKind > RT_SHAPE_MESH) return fail(RTGPU_ERR_UNSUPPORTED, "unknown shape kind");
        }
    }
    for (uint32_t i = 0; i < s->numTriangles; ++i)
        if (s->vertexIndices[i].materialIndex != RT_NO_MATERIAL && s->vertexIndices[i].materialIndex >= s->numMaterials) return fail(RTGPU_ERR_INVALID_ARGUMENT, "triangle material index out of range");
    for (uint32_t i = 0; i < s->numGlobalLights; ++i) if (s->globalLights[i] >= s->numLights) return fail(RTGPU_ERR_INVALID_ARGUMENT, "global light index out of range");
    for (uint32_t i = 0; i < s->numMaterials; ++i) if (s->materials[i].bsdf > RT_BSDF_ROUGH_PLASTIC) return fail(RTGPU_ERR_UNSUPPORTED, "unknown BSDF kind");
    if (s->numMaterials >= (1u << 22)) return fail(RTGPU_ERR_UNSUPPORTED, "more than 4M materials");   // the path flags hold a material index in 23 bits
    // textures: known kinds and formats, rows / blocks / palettes inside the texel blob, mixes nested at most one level deep
    for (uint32_t i = 0; i < s->numTextures; ++i)
    {
        const RtTexture& t = s->textures[i];
        if (t.kind == RT_TEXTURE_CHECKERBOARD || t.kind == RT_TEXTURE_CONST) continue;
        if (t.kind == RT_TEXTURE_NOISE) { if (t.numOctaves == 0 || t.numOctaves > 20) return fail(RTGPU_ERR_INVALID_ARGUMENT, "noise octaves must be 1..20"); continue; }
        if (t.kind == RT_TEXTURE_MIX)
        {
            const uint32_t children[3] = { t.mixA, t.mixB, t.mixWeight };
            for (uint32_t child : children)
            {
                if (child >= s->numTextures) return fail(RTGPU_ERR_INVALID_ARGUMENT, "mix texture child index out of range");
                const RtTexture& c = s->textures[child];
                if (c.kind != RT_TEXTURE_MIX) continue;
                const uint32_t grandChildren[3] = { c.mixA, c.mixB, c.mixWeight };
                for (uint32_t g : grandChildren)
                    if (g >= s->numTextures || s->textures[g].kind == RT_TEXTURE_MIX) return fail(RTGPU_ERR_UNSUPPORTED, "mix textures nested more than one level deep");
            }
            continue;
        }
        if (t.kind != RT_TEXTURE_BITMAP) return fail(RTGPU_ERR_UNSUPPORTED, "unknown texture kind");
        uint32_t bits = 0;
        switch (t.format)
        {
        case RT_FORMAT_R8_UNORM: case RT_FORMAT_B8G8R8A8_UNORM_PALETTE: case RT_FORMAT_BC5: bits = 8; break;
        case RT_FORMAT_R8G8_UNORM: case RT_FORMAT_R16_UNORM: case RT_FORMAT_R16_HALF: case RT_FORMAT_B5G6R5_UNORM: bits = 16; break;
        case RT_FORMAT_B8G8R8_UNORM: bits = 24; break;
        case RT_FORMAT_B8G8R8A8_UNORM: case RT_FORMAT_R8G8B8A8_UNORM: case RT_FORMAT_R16G16_UNORM: case RT_FORMAT_R32_FLOAT: case RT_FORMAT_R16G16_HALF:
        case RT_FORMAT_R11G11B10_FLOAT: case RT_FORMAT_R9G9B9E5_SHAREDEXP: bits = 32; break;
        case RT_FORMAT_R16G16B16_HALF: bits = 48; break;
        case RT_FORMAT_R16G16B16A16_UNORM: case RT_FORMAT_R32G32_FLOAT: case RT_FORMAT_R16G16B16A16_HALF: bits = 64; break;
        case RT_FORMAT_R32G32B32_FLOAT: bits = 96; break;
        case RT_FORMAT_R32G32B32A32_FLOAT: bits = 128; break;
        case RT_FORMAT_BC1: case RT_FORMAT_BC4: bits = 4; break;
        default: return fail(RTGPU_ERR_UNSUPPORTED, "unknown bitmap format");
        }
        if (t.width == 0 || t.height == 0 || t.width > 65536u || t.height > 65536u) return fail(RTGPU_ERR_INVALID_ARGUMENT, "invalid texture size");
        if (t.filter > RT_FILTER_BILINEAR_SMOOTHSTEP) return fail(RTGPU_ERR_INVALID_ARGUMENT, "unknown texture filter");
        const bool blocks = t.format == RT_FORMAT_BC1 || t.format == RT_FORMAT_BC4 || t.format == RT_FORMAT_BC5;
        uint64_t extent;
        if (blocks)
        {
            if ((t.width & 3u) || (t.height & 3u)) return fail(RTGPU_ERR_INVALID_ARGUMENT, "block-compressed textures need dimensions that are multiples of 4");
            extent = (uint64_t)(t.width / 4u) * (t.height / 4u) * (t.format == RT_FORMAT_BC5 ? 16u : 8u);
        }
        else
        {
            const uint32_t texelSize = bits / 8u;
            if (t.stride < t.width * texelSize) return fail(RTGPU_ERR_INVALID_ARGUMENT, "texture stride smaller than a row");
            extent = (uint64_t)t.stride * (t.height - 1u) + (uint64_t)t.width * texelSize;
        }
        if (!s->texelData || t.dataOffset + extent > s->texelBytes) return fail(RTGPU_ERR_INVALID_ARGUMENT, "texture data outside texelData");
        if (t.format == RT_FORMAT_B8G8R8A8_UNORM_PALETTE && t.paletteOffset + 1024u > s->texelBytes) return fail(RTGPU_ERR_INVALID_ARGUMENT, "texture palette (256 entries) outside texelData");
    }
    auto textureOk = [&](uint32_t index) { return index == RT_NO_TEXTURE || index < s->numTextures; };
    for (uint32_t i = 0; i < s->numMaterials; ++i)
    {
        const RtMaterial& m = s->materials[i];
        if (!textureOk(m.baseColorTexture) || !textureOk(m.emissionTexture) || !textureOk(m.roughnessTexture) || !textureOk(m.metalnessTexture) || !textureOk(m.normalMapTexture))
            return fail(RTGPU_ERR_INVALID_ARGUMENT, "material texture index out of range");
    }
    for (uint32_t i = 0; i < s->numLights; ++i) if (!textureOk(s->lights[i].texture)) return fail(RTGPU_ERR_INVALID_ARGUMENT, "light texture index out of range");

    freeScene(c);
    RtSceneDesc d = *s;
    int r;
    if ((r = uploadArray(c, s->topNodes, s->numTopNodes, &d.topNodes))) return r;
    if ((r = uploadArray(c, s->objects, s->numObjects, &d.objects))) return r;
    if ((r = uploadArray(c, s->lights, s->numLights, &d.lights))) return r;
    if ((r = uploadArray(c, s->globalLights, s->numGlobalLights, &d.globalLights))) return r;
    if ((r = uploadArray(c, s->materials, s->numMaterials, &d.materials))) return r;
    if ((r = uploadArray(c, s->meshes, s->numMeshes, &d.meshes))) return r;
    {
        // mesh trees go to the device in BREADTH-FIRST order (root at 0, node 1 unused, child pairs from 2 on as the reference lays them
        // out, levels one after the other): the same tree -- a node's childIndex is only a pointer -- with the top levels every ray
        // walks through contiguous at the front, which k_trace stages in LDS.  Leaves keep their triangle ranges.
        std::vector<RtNode> ordered(s->meshNodes, s->meshNodes + s->numMeshNodes);
        for (uint32_t m = 0; m < s->numMeshes; ++m)
        {
            const RtMesh& mesh = s->meshes[m];
            if (mesh.numNodes < 3u) continue;
            const RtNode* src = s->meshNodes + mesh.firstNode;
            RtNode* dst = ordered.data() + mesh.firstNode;
            std::vector<uint32_t> oldIndex; oldIndex.reserve(mesh.numNodes);   // oldIndex[new position]
            oldIndex.push_back(0u); oldIndex.push_back(1u);
            for (size_t k = 0; k < oldIndex.size() && oldIndex.size() + 2u <= mesh.numNodes; ++k)
            {
                if (k == 1u) continue;
                const RtNode& n = src[oldIndex[k]];
                if ((n.leaves & 0x3FFFFFFFu) != 0u) continue;
                dst[k] = n; dst[k].childIndex = (uint32_t)oldIndex.size();
                oldIndex.push_back(n.childIndex); oldIndex.push_back(n.childIndex + 1u);
            }
            for (size_t k = 0; k < oldIndex.size(); ++k) if (k != 1u && (src[oldIndex[k]].leaves & 0x3FFFFFFFu) != 0u) dst[k] = src[oldIndex[k]];
        }
        if ((r = uploadArray(c, ordered.data(), ordered.size(), &d.meshNodes))) return r;
    }
    if ((r = uploadArray(c, s->triangles, s->numTriangles, &d.triangles))) return r;
    {
        // de-indexed shading records (rt_device_core.h, TriangleShading), built once here
        std::vector<TriangleShading> records(s->numTriangles);
        if (!records.empty()) memset(records.data(), 0, records.size() * sizeof(TriangleShading));
        for (uint32_t m = 0; m < s->numMeshes; ++m)
        {
            const RtMesh& mesh = s->meshes[m];
            const RtVertexShading* vs = s->vertexShading + mesh.firstVertex;
            for (uint32_t t = 0; t < mesh.numTriangles; ++t)
            {
                const RtVertexIndices& idx = s->vertexIndices[mesh.firstTriangle + t];
                TriangleShading& out = records[mesh.firstTriangle + t];
                out.v[0] = vs[idx.i0]; out.v[1] = vs[idx.i1]; out.v[2] = vs[idx.i2];
                out.materialIndex = idx.materialIndex;
            }
        }
        const TriangleShading* dev = nullptr;
        if ((r = uploadArray(c, records.data(), records.size(), &dev))) return r;
        d.vertexIndices = reinterpret_cast<const RtVertexIndices*>(dev);
        d.vertexShading = nullptr;
    }
    if ((r = uploadArray(c, s->blueNoise, s->blueNoise ? (size_t)128 * 128 * 4 : 0, &d.blueNoise))) return r;
    if ((r = uploadArray(c, s->textures, s->numTextures, &d.textures))) return r;
    if ((r = uploadArray(c, s->texelData, s->numTextures ? (size_t)s->texelBytes : 0, &d.texelData))) return r;
    // single-mesh scenes (Scene::Traverse's one-object bypass): the re-encoded tree of the default traversal kernel
    memset(&c->wide, 0, sizeof(c->wide));
    memset(&c->wide2, 0, sizeof(c->wide2));
    const bool singleMesh = s->numObjects == 1u && s->objects[0].objectKind == RT_OBJECT_SHAPE && s->objects[0].shapeKind == RT_SHAPE_MESH;
    bool anyMesh = false;
    for (uint32_t o = 0; o < s->numObjects; ++o) anyMesh = anyMesh || (s->objects[o].objectKind == RT_OBJECT_SHAPE && s->objects[o].shapeKind == RT_SHAPE_MESH);
    // (a handful of analytic objects -- sphere + area light: a top-level tree of one or three nodes -- gain nothing from wider nodes and pay
    //  for the re-trace launch: measured 3-5 % slower, the binary kernel keeps them)
    if (!singleMesh && s->numObjects > 1u && (anyMesh || s->numTopNodes >= 7u))
    {
        // every other scene: the two-level 4-wide walk (rt_trace_wide2.inl).  One node / gate array for all levels; levels[o] for mesh object o,
        // levels[numObjects] for the top-level tree.  A level that cannot be built (a malformed tree) leaves the scene to the binary walk.
        std::vector<float4> allNodes, allGates;
        std::vector<WideLevel> levels(s->numObjects + 1u);
        memset(levels.data(), 0, levels.size() * sizeof(WideLevel));
        bool ok = true;
        auto append = [&](const WideLevelBuild& b, WideLevel& level, uint32_t triBase)
        {
            level.nodeBase = (uint32_t)(allNodes.size() / 4u); level.gateBase = (uint32_t)allGates.size(); level.triBase = triBase; level.valid = 1u;
            memcpy(level.base, b.base, sizeof(b.base)); memcpy(level.step, b.step, sizeof(b.step)); memcpy(level.bound, b.bound, sizeof(b.bound));
            allNodes.insert(allNodes.end(), b.nodes.begin(), b.nodes.end()); allGates.insert(allGates.end(), b.gate.begin(), b.gate.end());
        };
        if (s->numObjects > 1u)
        {
            const WideLevelBuild top = buildWideLevel(s->topNodes, s->numTopNodes, s->numObjects, topDepth);
            if (top.ok) append(top, levels[s->numObjects], 0u); else ok = false;
        }
        std::unordered_map<uint32_t, uint32_t> builtMesh;   // mesh index -> the first object whose level holds its tree (instances share it)
        for (uint32_t o = 0; o < s->numObjects && ok; ++o)
        {
            const RtObject& obj = s->objects[o];
            if (obj.objectKind != RT_OBJECT_SHAPE || obj.shapeKind != RT_SHAPE_MESH) continue;
            const RtMesh& mesh = s->meshes[obj.meshIndex];
            if (mesh.numNodes == 0u) continue;   // nothing to hit (Traverse_Object returns at once)
            const auto found = builtMesh.find(obj.meshIndex);
            if (found != builtMesh.end()) { levels[o] = levels[found->second]; continue; }
            const WideLevelBuild b = buildWideLevel(s->meshNodes + mesh.firstNode, mesh.numNodes, mesh.numTriangles, bvhDepth(s->meshNodes + mesh.firstNode, mesh.numNodes));
            if (!b.ok) { ok = false; break; }
            append(b, levels[o], mesh.firstTriangle);
            builtMesh[obj.meshIndex] = o;
        }
        if (ok && (allNodes.size() / 4u) < RT_NODE_CHILD_MASK && allGates.size() < 0xFFFFFFFFull)
        {
            const float4* devNodes = nullptr; const float4* devGates = nullptr; const WideLevel* devLevels = nullptr;
            if ((r = uploadArray(c, allNodes.data(), allNodes.size(), &devNodes))) return r;
            if ((r = uploadArray(c, allGates.data(), allGates.size(), &devGates))) return r;
            if ((r = uploadArray(c, levels.data(), levels.size(), &devLevels))) return r;
            c->wide2.nodes = devNodes; c->wide2.gate = devGates; c->wide2.levels = devLevels; c->wide2.numObjects = s->numObjects;
            c->walkNodeBytes[RTGPU_WALK_WIDE2] = allNodes.size() * sizeof(float4); c->walkLeafBoxBytes[RTGPU_WALK_WIDE2] = allGates.size() * sizeof(float4);
        }
    }
    if (singleMesh)
    {
        const RtMesh& mesh = s->meshes[s->objects[0].meshIndex];
        const QuantBuild q = buildQuantBvh(s->meshNodes + mesh.firstNode, mesh.numNodes, mesh.numTriangles, maxMeshDepth);
        if (q.ok)
        {
            const float4* devGate = nullptr;
            if ((r = uploadArray(c, q.gate.data(), q.gate.size(), &devGate))) return r;
            const WideBuild w = buildWideBvh(s->meshNodes + mesh.firstNode, mesh.numNodes, q);
            if (w.ok)
            {
                const float4* devWide = nullptr;
                if ((r = uploadArray(c, w.nodes.data(), w.nodes.size(), &devWide))) return r;
                c->wide.nodes = devWide; c->wide.gate = devGate; c->wide.numNodes = (uint32_t)(w.nodes.size() / 4u);
                c->walkNodeBytes[RTGPU_WALK_WIDE] = w.nodes.size() * sizeof(float4); c->walkLeafBoxBytes[RTGPU_WALK_WIDE] = q.gate.size() * sizeof(float4);
                memcpy(c->wide.base, q.base, sizeof(q.base)); memcpy(c->wide.step, q.step, sizeof(q.step)); memcpy(c->wide.bound, q.bound, sizeof(q.bound));
            }
        }
    }
    c->sceneDev = d;
    c->walkNodeBytes[RTGPU_WALK_BINARY] = ((uint64_t)s->numTopNodes + s->numMeshNodes) * sizeof(RtNode); c->walkTriangleBytes = (uint64_t)s->numTriangles * sizeof(RtTriangle);
    c->numLights = s->numLights;
    c->traversalStackNeed = topDepth + maxMeshDepth;
    bool lean = !(getenv("RTGPU_NO_LEAN") && atoi(getenv("RTGPU_NO_LEAN")) != 0);
    for (uint32_t i = 0; i < s->numObjects && lean; ++i) lean = s->objects[i].objectKind == RT_OBJECT_SHAPE && s->objects[i].shapeKind == RT_SHAPE_MESH;
    for (uint32_t i = 0; i < s->numMaterials && lean; ++i) lean = s->materials[i].bsdf == RT_BSDF_DIFFUSE;
    for (uint32_t i = 0; i < s->numLights && lean; ++i) lean = s->lights[i].type == RT_LIGHT_BACKGROUND || s->lights[i].type == RT_LIGHT_DIRECTIONAL;
    bool textured = false;
    for (uint32_t i = 0; i < s->numMaterials; ++i)
        textured = textured || (s->materials[i].baseColorTexture & s->materials[i].emissionTexture & s->materials[i].roughnessTexture & s->materials[i].metalnessTexture & s->materials[i].normalMapTexture) != RT_NO_TEXTURE;
    for (uint32_t i = 0; i < s->numLights; ++i) textured = textured || s->lights[i].texture != RT_NO_TEXTURE;
    c->leanScene = lean ? (textured ? 2 : 1) : (textured ? 0 : 3);
    {
        // class 4: a lean scene whose textures are all plain 8-bit BGR(A) / RGBA or half-float RGBA bitmaps (what Demo/MeshLoader.cpp makes of an OBJ's
        // diffuse and normal maps: 24-bit .bmp files) -- the shading kernel inlines their evaluation.  RTGPU_NO_SIMPLE_TEXTURES=1: class 2 instead.
        bool simple = c->leanScene == 2 && s->numTextures != 0u && !(getenv("RTGPU_NO_SIMPLE_TEXTURES") && atoi(getenv("RTGPU_NO_SIMPLE_TEXTURES")) != 0);
        for (uint32_t i = 0; i < s->numTextures && simple; ++i) simple = s->textures[i].kind == RT_TEXTURE_BITMAP && RT_FORMAT_IS_SIMPLE(s->textures[i].format);
        if (simple) c->leanScene = 4;
    }
    // a delta directional light whose direction has an exactly-zero component: EVERY next-event ray towards it is axis-parallel and goes through the re-trace launches (launchRetrace)
    c->axisParallelSun = false;
    for (uint32_t i = 0; i < s->numLights; ++i)
        if (s->lights[i].type == RT_LIGHT_DIRECTIONAL && s->lights[i].isDelta && (s->lights[i].transform[8] == 0.0f || s->lights[i].transform[9] == 0.0f || s->lights[i].transform[10] == 0.0f)) c->axisParallelSun = true;
    c->sceneReady = true;
    c->vcm.havePhotons = false;   // photons of another scene
    return RTGPU_OK;
}

// slot -> pixel table: owned 64x64 tiles (tile % worldSize == rank), 8x8 blocks inside a tile, so that a
// wave covers an 8x8 pixel block (coherent primary rays)
static std::vector<uint32_t> buildSlotTable(uint32_t width, uint32_t height, RtgpuShard shard, const std::vector<uint8_t>& activeMask)
{
    std::vector<uint32_t> slots;
    slots.reserve((size_t)width * height / (shard.worldSize ? shard.worldSize : 1) + 4096);
    const uint32_t tilesX = (width + 63u) / 64u, tilesY = (height + 63u) / 64u;
    for (uint32_t ty = 0; ty < tilesY; ++ty)
        for (uint32_t tx = 0; tx < tilesX; ++tx)
        {
            const uint32_t tile = ty * tilesX + tx;
            if (shard.worldSize > 1 && tile % shard.worldSize != shard.rank) continue;
            for (uint32_t by = 0; by < 8; ++by)
                for (uint32_t bx = 0; bx < 8; ++bx)
                    for (uint32_t py = 0; py < 8; ++py)
                        for (uint32_t px = 0; px < 8; ++px)
                        {
                            const uint32_t x = tx * 64 + bx * 8 + px, y = ty * 64 + by * 8 + py;
                            if (x < width && y < height && (activeMask.empty() || activeMask[(size_t)y * width + x])) slots.push_back(x | (y << 16));
                        }
        }
    return slots;
}

// slot -> pixel table of the pixels this context renders: owned tiles, active blocks
static int rebuildSlots(RtgpuContext* c)
{
    if (c->slotPixel) { (void)hipFree(c->slotPixel); c->slotPixel = nullptr; }
    const std::vector<uint32_t> slots = buildSlotTable(c->width, c->height, c->shard, c->activeMask);
    c->numSlots = (uint32_t)slots.size();
    // launches of a batch should stay large enough to fill 256 CUs: a 1/8 shard of a 1080p frame batches 16 passes
    // (measured on 1/8 of the Sponza-class frame: 0.57 -> 0.50 ms per pass), a full frame 8
    if (!c->passBatchFromEnv)
    {
        static const uint32_t streamBase = getenv("RTGPU_PASS_BATCH_BASE") ? (uint32_t)atoi(getenv("RTGPU_PASS_BATCH_BASE")) : 5u;
        // (small frames / 1/8 shards of a full-HD frame: 20 passes per batch -- 16 -> 20: +3 % at the driver's 20 steps, profiles/r04_tail_sweep.txt)
        static const uint32_t smallBatch = getenv("RTGPU_SMALL_FRAME_BATCH") ? (uint32_t)atoi(getenv("RTGPU_SMALL_FRAME_BATCH")) : 20u;
        c->passBatch = c->numSlots != 0 && c->numSlots < 400000u ? (smallBatch ? smallBatch : 1u) : (streamBase ? streamBase : 1u);   // full frames: 5 -> 10 -> 20 -> 24 while streaming, one size per round of the lanes (flushBatch)
        // (very large frames: fewer passes per launch, an arena of 8 passes of an 8K frame would be 47 GB)
        while (c->passBatch > 1u && (size_t)c->numSlots * c->passBatch * ((size_t)R_NUM_BASE + RT_SHADOW_RECORDS) * sizeof(float4) > ((size_t)24 << 30)) c->passBatch /= 2u;
    }
    c->passBatchBase = c->passBatch;
    // four batch lanes: with the short re-trace launches behind k_trace_wide and streams that start with small batches, a fourth
    // concurrent launch sequence pays on full frames too (20 passes between read-backs: +4.6 %; 64 and 256 passes: unchanged)
    if (!c->lanesChosen) { c->numLanes = 4u; if (c->nextLane >= c->numLanes) c->nextLane = 0; }
    if (c->numSlots)
    {
        HIP_TRY(hipMalloc((void**)&c->slotPixel, slots.size() * sizeof(uint32_t)));
        HIP_TRY(rtMemcpy(c->slotPixel, slots.data(), slots.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return RTGPU_OK;
}

static int rebuildFilm(RtgpuContext* c)
{
    freeFilm(c);
    if (c->width == 0 || c->height == 0) return RTGPU_OK;
    const size_t n = (size_t)c->width * c->height * 3;
    HIP_TRY(hipMalloc((void**)&c->sum, n * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&c->secondary, n * sizeof(float)));
    HIP_TRY(hipMemset(c->sum, 0, n * sizeof(float)));
    HIP_TRY(hipMemset(c->secondary, 0, n * sizeof(float)));
    HIP_TRY(hipStreamSynchronize(nullptr));   // the batch lanes are non-blocking streams: they do not wait for the null stream's memsets
    c->activeMask.clear();   // a new film starts with the whole image active
    c->vcm.havePhotons = false;   // recorded per slot of the old film
    return rebuildSlots(c);
}

RTGPU_API int rtgpu_resize(RtgpuContext* c, uint32_t width, uint32_t height)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    if (width == 0 || height == 0 || width > 65536u || height > 65536u) return fail(RTGPU_ERR_INVALID_ARGUMENT, "Invalid viewport size");
    RT_FAN_OUT(c, rtgpu_resize(peer, width, height));
    HIP_TRY(hipSetDevice(c->device));
    { int fr = flushPending(c); if (fr) return fr; }
    HIP_TRY(syncLanes(c));
    c->width = width; c->height = height;
    return rebuildFilm(c);
}

RTGPU_API int rtgpu_set_shard(RtgpuContext* c, RtgpuShard shard)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    if (shard.worldSize == 0 || shard.rank >= shard.worldSize) return fail(RTGPU_ERR_INVALID_ARGUMENT, "invalid shard");
    if (!c->peers.empty() || c->isPeer) return fail(RTGPU_ERR_UNSUPPORTED, "a multi-device context shards the frame itself (rtgpu_create_multi)");
    HIP_TRY(hipSetDevice(c->device));
    { int fr = flushPending(c); if (fr) return fr; }
    HIP_TRY(syncLanes(c));
    c->shard = shard;
    return rebuildFilm(c);
}

RTGPU_API int rtgpu_reset(RtgpuContext* c)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    RT_FAN_OUT(c, rtgpu_reset(peer));
    HIP_TRY(hipSetDevice(c->device));
    { int fr = flushPending(c); if (fr) return fr; }
    HIP_TRY(syncLanes(c));
    if (c->sum)
    {
        const size_t n = (size_t)c->width * c->height * 3;
        HIP_TRY(hipMemset(c->sum, 0, n * sizeof(float)));
        HIP_TRY(hipMemset(c->secondary, 0, n * sizeof(float)));
    }
    HIP_TRY(hipMemset(c->counters, 0, 16 * sizeof(unsigned long long)));
    HIP_TRY(hipStreamSynchronize(nullptr));   // (see rebuildFilm)
    int r = resolveTimed(c); if (r) return r;
    memset(c->kernelMs, 0, sizeof(c->kernelMs)); memset(c->kernelLaunches, 0, sizeof(c->kernelLaunches));
    return RTGPU_OK;
}

// Streaming grows the batch 8 -> 16 -> 24 passes; frames beyond full HD stop earlier so that an arena stays below ~24 GB
// (176 bytes per slot with one NEE request per vertex: 4K frames reach 16 passes, 8K frames stay at 8 and below)
static uint32_t maxStreamingBatch(const RtgpuContext* c)
{
    const size_t perPass = (size_t)(c->numSlots ? c->numSlots : 1) * ((size_t)R_NUM_BASE + RT_SHADOW_RECORDS) * sizeof(float4);
    static const uint32_t most = getenv("RTGPU_MAX_STREAM_BATCH") ? (uint32_t)atoi(getenv("RTGPU_MAX_STREAM_BATCH")) : 24u;   // tuning knob (a multiple of 8, at most 64)
    uint32_t batch = most < 8u ? 8u : (most > RT_SEED_RING / 2 ? RT_SEED_RING / 2 : most);
    while (batch > 8u && perPass * batch > ((size_t)24 << 30)) batch -= 8u;
    return batch;
}

// Device bytes one path slot costs a batch lane when a vertex can have `maxLights` next-event requests: the records of both arenas
// (the second one only exists with dense path state: up to RT_DENSE_MAX_LIGHTS requests per vertex), the parked radiance, the queues.  LightSamplingStrategy::All with many
// lights makes slots fat (64 lights: 2.2 KB), so the batch a lane can hold shrinks with it -- down to one pass.
static size_t bytesPerSlot(uint32_t maxLights)
{
    if (maxLights == 0) maxLights = 1;
    const size_t arena = ((size_t)R_NUM_BASE + (size_t)maxLights * RT_SHADOW_RECORDS) * sizeof(float4);
    return arena * (maxLights <= RT_DENSE_MAX_LIGHTS ? 2u : 1u) + sizeof(float4) + sizeof(uint32_t) * (3u + 3u * (size_t)maxLights);
}
static uint32_t maxBatchFor(const RtgpuContext* c, uint32_t maxLights)
{
    const size_t perPass = (size_t)(c->numSlots ? c->numSlots : 1) * bytesPerSlot(maxLights);
    const size_t batch = c->laneBudgetBytes / perPass;
    return batch < 1u ? 1u : (batch > RT_SEED_RING / 2 ? RT_SEED_RING / 2 : (uint32_t)batch);
}
// slots an arena is allocated for: the regions of dense path state need a margin each (a region's share of a launch is only
// roughly a sixteenth: blocks take turns)
static size_t arenaCapacityFor(size_t slots) { return (size_t)RT_DENSE_SHARDS * ((slots + RT_DENSE_SHARDS - 1u) / RT_DENSE_SHARDS + 65536u); }

static int ensurePaths(RtgpuContext* c, BatchLane& l, uint32_t maxLights, uint32_t maxDepth)
{
    if (maxLights == 0) maxLights = 1;
    const bool wantDense = c->denseAllowed && maxLights <= RT_DENSE_MAX_LIGHTS;
    for (int attempt = 0; attempt < 2; ++attempt)
    {
        uint32_t maxBatch = (c->passBatchFromEnv || c->numSlots < 400000u) ? c->passBatch : maxStreamingBatch(c);   // the largest batch streaming can reach
        if (maxBatch > maxBatchFor(c, maxLights)) maxBatch = maxBatchFor(c, maxLights);
        const size_t wanted = (size_t)(c->numSlots ? c->numSlots : 1) * maxBatch;
        if (l.paths.base && l.paths.capacity >= arenaCapacityFor(wanted) && l.paths.maxLights >= maxLights && (!wantDense || (l.paths2.base && l.homeCapacity >= wanted))) break;
        HIP_TRY(hipStreamSynchronize(l.stream));
        freePaths(l);
        if (attempt == 0)
        {
            // The lane budget of rtgpu_create is a guess made before anything was allocated.  Contexts that share a device (several
            // renderers in one process, rtgpu_create_multi with a repeated index) see less: what is free NOW is shared by the lanes of
            // this context that still have to allocate, and the batch a lane may hold shrinks with it instead of a late out-of-memory.
            size_t freeBytes = 0, totalBytes = 0;
            if (hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess)
            {
                uint32_t lanesLeft = 0;
                for (uint32_t i = 0; i < c->numLanes; ++i) if (!c->lanes[i].paths.base) lanesLeft++;
                const size_t share = (size_t)((double)freeBytes * 0.9) / (lanesLeft ? lanesLeft : 1u);
                if (share < c->laneBudgetBytes)
                {
                    c->laneBudgetBytes = share;   // size the arenas again under the smaller budget
                    if (c->passBatch > maxBatchFor(c, maxLights)) c->passBatch = maxBatchFor(c, maxLights);
                    if (c->passBatchBase > c->passBatch) c->passBatchBase = c->passBatch;
                    continue;
                }
            }
        }
        const size_t cap = arenaCapacityFor(wanted);
        if (cap >= 0xFFFFFFFFull) return fail(RTGPU_ERR_UNSUPPORTED, "pixels x pass batch exceeds the slot index range");
        const size_t records = ((size_t)R_NUM_BASE + (size_t)maxLights * RT_SHADOW_RECORDS) * cap;
        HIP_TRY(hipMalloc((void**)&l.paths.base, records * sizeof(float4)));
        HIP_TRY(hipMalloc((void**)&l.queues[0], cap * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&l.queues[1], cap * sizeof(uint32_t)));
        if ((unsigned long long)cap * maxLights >= 0xFFFFFFFFull) return fail(RTGPU_ERR_UNSUPPORTED, "pixels x lights exceeds the NEE request index range");
        HIP_TRY(hipMalloc((void**)&l.shadowQueues[0], cap * maxLights * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&l.shadowQueues[1], cap * maxLights * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&l.exactQueue, cap * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&l.exactShadowQueue, cap * maxLights * sizeof(uint32_t)));
        l.paths.capacity = (uint32_t)cap; l.paths.maxLights = maxLights;
        if (wantDense)
        {
            HIP_TRY(hipMalloc((void**)&l.paths2.base, records * sizeof(float4)));
            HIP_TRY(hipMalloc((void**)&l.home, wanted * sizeof(float4)));
            l.homeCapacity = wanted;
            l.paths2.capacity = (uint32_t)cap; l.paths2.maxLights = maxLights;
        }
        break;
    }
    if (l.queueCountCapacity < maxDepth + 2)
    {
        HIP_TRY(hipStreamSynchronize(l.stream));
        if (l.queueCounts) (void)hipFree(l.queueCounts);
        if (l.denseCounts) (void)hipFree(l.denseCounts);
        l.queueCountCapacity = maxDepth + 2;
        HIP_TRY(hipMalloc((void**)&l.queueCounts, (size_t)8 * l.queueCountCapacity * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&l.denseCounts, (size_t)2 * RT_DENSE_SHARDS * (l.queueCountCapacity + 1u) * sizeof(uint32_t)));
    }
    return RTGPU_OK;
}

// The 4-wide tree: single-mesh scenes, intersection counters off (they belong to the reference's walk).  Stack: 24 entries per lane, a
// ray that would need more goes to the binary-tree kernel.
static bool useWide(const RtgpuContext* c) { return (c->wide.nodes != nullptr || (c->wide2.nodes != nullptr && c->wide2Allowed)) && c->wideAllowed && !c->countIntersections; }

static void launchTraceWide(RtgpuContext* c, hipStream_t stream, const Paths& paths, const uint32_t* tq, const uint32_t* tqc, const uint32_t* tsq, const uint32_t* tsc,
                            uint32_t* cursor, uint32_t* exactQueue, uint32_t* exactCount, uint32_t* exactShadowQueue, uint32_t* exactShadowCount, float shadowOffset,
                            const uint32_t* denseCounts, uint32_t denseShardCapacity, bool mayTraceUndecidedRaysItself = true, uint32_t bounce = 0u)
{
    // A block traces the rays its walk does not decide itself (rt_trace_wide.inl) where launches are short: a 1/8 shard of a full-HD frame gains 10 %
    // (ten launches per batch less to wait for), a full frame loses 1.4 % (a block holds its slot of the CU while one wave walks; the separate launch
    // ran beside the other lanes' kernels) -- profiles/r04_local_exact_ab.txt.  RTGPU_LOCAL_EXACT=0 / 1 forces it.
    static const int localExactEnv = getenv("RTGPU_LOCAL_EXACT") ? atoi(getenv("RTGPU_LOCAL_EXACT")) : -1;
    // (larger frames: from this bounce on -- the late launches of a batch are short whatever the frame; 255 = never)
    static const uint32_t localExactFromBounce = getenv("RTGPU_LOCAL_EXACT_FROM") ? (uint32_t)atoi(getenv("RTGPU_LOCAL_EXACT_FROM")) : 255u;
    // (the second walk runs on the kernel's 24-entry stacks: scenes whose binary trees need deeper ones keep the separate launch)
    // (never in front of the bidirectional integrator: its light paths produce degenerate closest-hit rays -- an emitted direction that is exactly a coordinate
    //  axis -- which walk alone for milliseconds and would hold a whole block's slot of the CU meanwhile: 16.5 -> 26 ms per pass, profiles/r04_vcm_wide_ab.txt, measured
    //  when the separate launch still handed them on to k_trace_monster; that hand-over is opt-in since round 5 (launchRetrace), the separate launch stays: it
    //  holds one block per CU instead of the traversal grid)
    const bool localExact = mayTraceUndecidedRaysItself && c->traversalStackNeed <= 24u && (localExactEnv >= 0 ? localExactEnv != 0 : (c->localRetrace >= 0 ? c->localRetrace != 0 : (c->numSlots < 400000u || bounce >= localExactFromBounce)));   // (round 5, with re-trace launches that hand long rays on and share subtrees early: a 1/8 shard still gains 3 % from it, a 1/4 shard (518 k pixels) now LOSES 2 %, halves 0: profiles/r05_shard_policy.txt)
    static const uint32_t chunkMin = getenv("RTGPU_WIDE_CHUNK_MIN") ? (uint32_t)atoi(getenv("RTGPU_WIDE_CHUNK_MIN")) : 64u;   // tuning knob
    WideTuning tune = { c->tune.refillMinIdle, c->tune.otherMinLanes, shadowOffset, exactQueue, exactCount, exactShadowQueue, exactShadowCount, denseCounts, denseShardCapacity,
                        chunkMin < 64u ? 64u : chunkMin, localExact ? 1u : 0u, 0u };
    // test hook, read per launch: a wave whose work queue ran dry N loop iterations ago hands the rays it still walks -- hits half found, written through -- to the
    // re-trace launch (the stack-overflow path, which the benchmark frame never takes).  As a schedule it moves time, it does not save any: what k_trace_wide's drain
    // loses (-5.6 % at N = 8) the re-trace launches gain, with or without k_trace_monster behind them (profiles/r05_drain_abort_ab.txt)
    if (const char* e = getenv("RTGPU_WIDE_DRAIN_ABORT")) tune.drainAbortAfter = (uint32_t)atoi(e);
    // The order the work queue {closest-hit rays of bounce k, any-hit requests of bounce k - 1} is taken in: a launch ends with the drain of its last rays, so the
    // SHORT rays belong at the end.  Round 5 took it from its END (any-hit requests first): unoccluded next-event rays, which no hit ever shortens, were the long
    // ones (trace -1 %, shards +2 %, profiles/r05_claim_order_ab.txt).  Round 6's far-first order made any-hit rays the short ones (9.9 interior visits against a
    // closest-hit ray's 17), and the queue is taken front to back again: trace 47.5 -> 45.5 ms per 25 passes, +1 % at 256 passes, +2 % on a 1/8 shard
    // (profiles/r06_claim_order_ab.txt).  RTGPU_WIDE_REVERSE=1: from the end (read per launch: the tests run both orders)
    tune.reverseOrder = 0u;
    if (const char* e = getenv("RTGPU_WIDE_REVERSE")) tune.reverseOrder = (uint32_t)atoi(e);
    // any-hit rays walk the FARTHEST child they enter first (rt_trace_wide.inl: occlusion is an OR over the candidates, and the occluders of a ray that starts on a
    // surface are far from it); RTGPU_ANYHIT_FAR_FIRST=0: nearest first like closest-hit rays (read per launch: the tests run both orders)
    tune.anyHitFarFirst = 1u;
    if (const char* e = getenv("RTGPU_ANYHIT_FAR_FIRST")) tune.anyHitFarFirst = (uint32_t)atoi(e);
    const dim3 grid(c->numCUs * (c->travBlocksPerCU ? c->travBlocksPerCU : 5u)), block(RT_BLOCK);
    LaunchTimer t(c, stream, KC_TRACE);
    if (c->wide.nodes == nullptr)
    {
        // a two-level scene (rt_trace_wide2.inl): held to five waves per SIMD (110 -> 96 VGPRs, 8 bytes of scratch: Cornell box trace -7 %, +2 % end to
        // end), 30 KB of LDS per block
        const dim3 grid2(c->numCUs * (c->travBlocksPerCU ? c->travBlocksPerCU : 5u));
        hipLaunchKernelGGL((k_trace_wide2<24>), grid2, block, 0, stream, c->sceneDev, c->wide2, paths, tq, tqc, tsq, tsc, cursor, c->counters, tune);
        return;
    }
    static const bool diag = getenv("RTGPU_WIDE_DIAG") != nullptr;       // walk statistics in the spare counters (tools/wide_diag.py)
    // the camera rays of a dense batch walk the tree as packets (rt_trace_packet.inl: a wave = an 8 x 8 pixel block, the node is uniform); RTGPU_PACKET=0: off
    const char* const packetEnv = getenv("RTGPU_PACKET");   // (read per launch: the tests switch it)
    const bool packets = !(packetEnv && atoi(packetEnv) == 0);
    if (packets && !diag && bounce == 0u && denseCounts != nullptr && tsq == nullptr && tq == nullptr)
    {
        static const uint32_t packetBlocksPerCU = getenv("RTGPU_PACKET_BLOCKS") ? (uint32_t)atoi(getenv("RTGPU_PACKET_BLOCKS")) : 8u;
        hipLaunchKernelGGL(k_trace_packet, dim3(c->numCUs * packetBlocksPerCU), block, 0, stream, c->sceneDev, c->wide, paths, cursor, c->counters, tune);
        return;
    }
    if (diag) tune.localExact = (uint32_t)atoi(getenv("RTGPU_WIDE_DIAG"));   // 2: stack-depth histogram instead of the visit counts (tools/wide_diag.py)
    if (diag) hipLaunchKernelGGL((k_trace_wide<24, true, false>), grid, block, 0, stream, c->sceneDev, c->wide, paths, tq, tqc, tsq, tsc, cursor, c->counters, tune);
    else if (localExact) hipLaunchKernelGGL((k_trace_wide<24, false, true>), grid, block, 0, stream, c->sceneDev, c->wide, paths, tq, tqc, tsq, tsc, cursor, c->counters, tune);
    else hipLaunchKernelGGL((k_trace_wide<24, false, false>), grid, block, 0, stream, c->sceneDev, c->wide, paths, tq, tqc, tsq, tsc, cursor, c->counters, tune);
}

// The re-trace launch behind a 4-wide walk: the reference's own walk (k_trace) over the rays the walk handed over (0.1 % of a launch), and -- single-mesh
// scenes -- k_trace_monster behind it for the closest-hit rays among them that k_trace gave up on: a direction that is exactly a coordinate axis turns
// two of three slab tests into inf - inf and the ray walks most of the tree, alone in its wave (1.0-1.6 ms launches at bounce 1 where an ordinary one
// takes 0.1-0.2 ms, profiles/r04_timeline_serial.txt); a whole block finds the same hit cooperatively.  `overflowQueue`: a queue of the lane nobody
// uses during this bounce's trace (dense path state: none of the slot queues is in use; slot-per-pixel: the one the next shade will fill).
static void launchRetrace(RtgpuContext* c, BatchLane& l, hipStream_t stream, const Paths& paths, uint32_t depth, uint32_t stackClass, uint32_t* overflowQueue)
{
    uint32_t* exactCounts = l.queueCounts + 4 * l.queueCountCapacity;
    uint32_t* exactShadowCounts = l.queueCounts + 5 * l.queueCountCapacity;
    uint32_t* exactCursors = l.queueCounts + 6 * l.queueCountCapacity;
    uint32_t* overflowCounts = l.queueCounts + 7 * l.queueCountCapacity;
    const char* const abortText = getenv("RTGPU_ABORT_RETRACE_AFTER");   // test hook, read per launch (0: every closest-hit ray in flight when its wave's queue runs dry goes to k_trace_monster)
    const int abortEnv = abortText ? atoi(abortText) : -1;
    // OFF by default since the axis-parallel prune (boxNearDegenerateAxes) made the rays it was built for short: on the benchmark frame no ray is handed over any more,
    // and the EMPTY k_trace_monster launch behind every re-trace launch is not free under concurrency -- its 64 blocks of 512 threads / 33 KB LDS wait for CU slots that the other
    // lanes' persistent traversal kernels hold: 27.7 ms summed over the 40 launches of the driver's timed region (profiles/r05_concurrency.txt), 2 % end to end
    // (profiles/r05_monsters_under_concurrency_ab.txt).  RTGPU_RETRACE_MONSTERS=1 (or the test hook RTGPU_ABORT_RETRACE_AFTER) switches the hand-over on; read per launch.
    const char* const monstersText = getenv("RTGPU_RETRACE_MONSTERS");
    const bool monstersWanted = monstersText ? atoi(monstersText) != 0 : abortText != nullptr;
    const bool monsters = monstersWanted && overflowQueue != nullptr && c->wide.nodes != nullptr && c->sceneDev.numObjects == 1u && !c->countIntersections;
    TravTuning exactTune = c->tune;
    exactTune.overflowQueue = monsters ? overflowQueue : nullptr; exactTune.overflowCount = monsters ? overflowCounts + depth : nullptr;
    exactTune.abortClosestAfter = abortEnv >= 0 ? (uint32_t)abortEnv : RT_ABORT_RETRACE_AFTER;
    exactTune.denseCounts = nullptr; exactTune.denseShardCapacity = 0u;
    // a re-trace launch's queue is dry after the first claim and its duration is its longest ray: an any-hit ray that slides along a wall it started on (a sun in a
    // coordinate plane: the ray lies IN the wall's plane, Moeller-Trumbore never accepts the coplanar triangles) walks ~170 nodes = 250 us alone.  Idle lanes take its
    // deferred subtrees after RT_RETRACE_SPLIT_AFTER drain iterations instead of the 32 of a full launch.
    static const uint32_t splitEnv = getenv("RTGPU_RETRACE_SPLIT_AFTER") ? (uint32_t)atoi(getenv("RTGPU_RETRACE_SPLIT_AFTER")) : 0u;   // tuning knob
    exactTune.splitAfter = splitEnv ? splitEnv : RT_RETRACE_SPLIT_AFTER;
    LaunchTimer t(c, stream, KC_RETRACE);
    // one block per CU serves the usual few thousand requests; above 1024 requests per CU (exactTune.fullGridAbove = numCUs * 1024) the whole traversal grid works
    // (decided on the device from the counts)
    // (only where the scene can produce such queues -- a delta sun with an exactly-zero direction component, c->axisParallelSun: the 1024 extra blocks that read two
    //  counts and leave cost an ordinary scene ~0.5 % end to end, profiles/r05_retrace_grid_ab.txt; RTGPU_RETRACE_FULL_GRID=0 / 1 forces it)
    static const int gridEnv = getenv("RTGPU_RETRACE_FULL_GRID") ? atoi(getenv("RTGPU_RETRACE_FULL_GRID")) : -1;
    const bool adaptiveGrid = gridEnv >= 0 ? gridEnv != 0 : c->axisParallelSun;
    const uint32_t fullBlocks = c->numCUs * (c->travBlocksPerCU ? c->travBlocksPerCU : (stackClass == 24u ? 5u : (stackClass == 32u ? 4u : 2u)));
    exactTune.baseBlocks = adaptiveGrid ? c->numCUs : 0u; exactTune.fullGridAbove = c->numCUs * 256u * 4u;
    const dim3 retraceGrid(adaptiveGrid ? fullBlocks : c->numCUs), block(RT_BLOCK);
#define RT_LAUNCH_RETRACE(S) hipLaunchKernelGGL((k_trace<S, false>), retraceGrid, block, 0, stream, c->sceneDev, paths, l.exactQueue, exactCounts + depth, l.exactShadowQueue, exactShadowCounts + depth, exactCursors + depth, c->counters, exactTune)
    if (stackClass == 24u) RT_LAUNCH_RETRACE(24); else if (stackClass == 32u) RT_LAUNCH_RETRACE(32); else RT_LAUNCH_RETRACE(64);
#undef RT_LAUNCH_RETRACE
    if (monsters) hipLaunchKernelGGL(k_trace_monster, dim3(64), dim3(RT_MONSTER_BLOCK), 0, stream, c->sceneDev, paths, overflowQueue, overflowCounts + depth);
}

// The bounce at which a dense batch hands its remaining paths to k_tail (0: never).  RTGPU_TAIL_DEPTH=n forces bounce n (0: off).
static uint32_t tailDepthFor(const RtgpuContext* c, uint32_t totalSlots, uint32_t maxRayDepth, bool denseAll, uint32_t stackClass)
{
    static const int env = getenv("RTGPU_TAIL_DEPTH") ? atoi(getenv("RTGPU_TAIL_DEPTH")) : -1;
    if (env == 0 || c->tailBounce == 0 || denseAll || c->wide.nodes == nullptr || !useWide(c) || stackClass != 24u || c->debugMode >= 0) return 0u;
    // Measured (profiles/r04_tail_sweep.txt, 20 passes): a 1/8 shard of the full-HD benchmark frame gains 6-8 % with the hand-over at bounce 4 or 5 (0.580 ->
    // 0.544 ms per pass, with 20-pass batches 0.575-0.606 -> 0.526-0.558; bounce 2: -20 %, 3: 0), a 1/4 shard +2 % at bounce 5 and +4 % at bounce 6 together with the block-local re-trace, halves and full frames lose 1-5 % at any bounce: the block-local
    // rounds pay a drain each and only beat the launch sequence where that is all floors.  So: small frames only.  Round 5 (faster traversal and re-trace launches,
    // profiles/r05_shard_policy.txt): bounce 6 beats 5 on the 1/8 shard too (0.499 -> 0.488 ms per pass), 4 loses everywhere, halves gain 0.6 % at 6 (left off).
    uint32_t depth = env > 0 ? (uint32_t)env : (c->tailBounce > 0 ? (uint32_t)c->tailBounce : (c->numSlots < 700000u ? 6u : 0u));
    (void)totalSlots;
    if (depth > maxRayDepth + 1u) return 0u;
    return depth;
}

// Submits the queued passes as one batch: generate -> {trace -> shade} per bounce -> trace -> accumulate.
static int flushBatch(RtgpuContext* c, uint32_t maxPasses)
{
    if (c->pending.empty()) return RTGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    uint32_t numPasses = maxPasses && maxPasses < c->pending.size() ? maxPasses : (uint32_t)c->pending.size();
    const DevPass& first = c->pending[0].pass;
    const uint32_t maxLights = first.lightSamplingStrategy == RT_LIGHT_SAMPLING_ALL ? c->numLights : 1u;
    BatchLane& l = c->lanes[c->nextLane];
    const int laneIndex = (int)c->nextLane;
    c->nextLane = (c->nextLane + 1u) % c->numLanes;
    // all lanes get their arenas with the first batch: a 3 GB hipMalloc costs tens of milliseconds
    int r = RTGPU_OK;
    for (uint32_t i = 0; i < c->numLanes && r == RTGPU_OK; ++i) r = ensurePaths(c, c->lanes[i], maxLights, first.maxRayDepth);
    if (r) { c->pending.clear(); return r; }
    // The arenas may have been sized under a budget that shrank at allocation (several contexts on one device, little free memory): the batch
    // is what the lane's allocation holds, the rest stays queued for the next flush.
    {
        const size_t perPass = c->numSlots ? c->numSlots : 1u;
        while (numPasses > 1u && (arenaCapacityFor(perPass * numPasses) > l.paths.capacity || (l.paths2.base && perPass * numPasses > l.homeCapacity))) --numPasses;
        if (arenaCapacityFor(perPass * numPasses) > l.paths.capacity || (l.paths2.base && perPass * numPasses > l.homeCapacity))
        {
            c->pending.clear();
            return fail(RTGPU_ERR_OUT_OF_MEMORY, "a batch lane's path-state arena does not hold one pass of this frame");
        }
    }

    // contiguous ring slots for the batch (seeds + pass constants); wait until their previous users have finished
    if (c->seedCursor + numPasses > RT_SEED_RING) c->seedCursor = 0;
    const uint32_t firstSlot = c->seedCursor; c->seedCursor = (c->seedCursor + numPasses) % RT_SEED_RING;
    for (uint32_t i = 0; i < numPasses; ++i)
    {
        const uint32_t slot = firstSlot + i;
        if (c->seedEventUsed[slot]) HIP_TRY(hipEventSynchronize(c->seedEvents[slot]));
        uint32_t* seedHost = c->seedRingHost + (size_t)slot * RTGPU_MAX_DIMENSIONS;
        uint32_t* seedDev = c->seedRingDev + (size_t)slot * RTGPU_MAX_DIMENSIONS;
        CtxPending& pd = c->pending[i];
        if (!pd.seeds.empty()) memcpy(seedHost, pd.seeds.data(), pd.seeds.size() * sizeof(uint32_t));
        pd.pass.seed = seedDev;
        c->passRingHost[slot] = pd.pass;
    }
    // the batch's ring slots are contiguous: ONE copy for the seeds of all its passes and one for their constants (a copy per pass in front of a 20-pass
    // batch of a small frame was 0.25 ms of stream time before the first kernel, profiles/r04_timeline_serial_shard8.txt)
    HIP_TRY(hipMemcpyAsync(c->seedRingDev + (size_t)firstSlot * RTGPU_MAX_DIMENSIONS, c->seedRingHost + (size_t)firstSlot * RTGPU_MAX_DIMENSIONS,
                           (size_t)numPasses * RTGPU_MAX_DIMENSIONS * sizeof(uint32_t), hipMemcpyHostToDevice, l.stream));
    HIP_TRY(hipMemcpyAsync(c->passRingDev + firstSlot, c->passRingHost + firstSlot, numPasses * sizeof(DevPass), hipMemcpyHostToDevice, l.stream));
    const DevPass* passesDev = c->passRingDev + firstSlot;

    const uint32_t totalSlots = c->numSlots * numPasses;
    static const uint32_t shadeBlocksPerCU = getenv("RTGPU_SHADE_BLOCKS_PER_CU") ? (uint32_t)atoi(getenv("RTGPU_SHADE_BLOCKS_PER_CU")) : 8u;   // tuning knob
    const uint32_t maxBlocks = c->numCUs * shadeBlocksPerCU;
    const uint32_t blocksNeeded = (totalSlots + RT_BLOCK - 1) / RT_BLOCK;
    const uint32_t pixelBlocks = (c->numSlots + RT_BLOCK - 1) / RT_BLOCK;
    const dim3 grid(blocksNeeded < maxBlocks ? blocksNeeded : maxBlocks), pixelGrid(pixelBlocks < maxBlocks ? pixelBlocks : maxBlocks), block(RT_BLOCK);

    // persistent traversal grids: enough resident waves to cover the latency of dependent node fetches; surplus
    // blocks simply queue (there is no inter-block dependency, only the atomic cursor)
    // LDS stack capacity in entries per lane: 24 (6 blocks per CU), 32 (4-5) or 64 (2); the scene's BVH depth decides
    const uint32_t stackClass = c->traversalStackNeed <= 24 ? 24u : (c->traversalStackNeed <= 32 ? 32u : 64u);
    const dim3 travGrid(c->numCUs * (c->travBlocksPerCU ? c->travBlocksPerCU : (stackClass == 24u ? 5u : (stackClass == 32u ? 4u : 2u))));
    uint32_t* pathCounts = l.queueCounts;
    uint32_t* shadowCounts = l.queueCounts + l.queueCountCapacity;
    uint32_t* cursors = l.queueCounts + 2 * l.queueCountCapacity;
    const uint32_t maxRayDepth = first.maxRayDepth;

    HIP_TRY(hipMemsetAsync(l.queueCounts, 0, (size_t)8 * l.queueCountCapacity * sizeof(uint32_t), l.stream));
    // DENSE path state (rt_dense.inl): one next-event request per vertex (LightSamplingStrategy::Single, or none: "Path Tracer"), or one per light
    // under LightSamplingStrategy::All with a handful of lights (the benchmark scene has two)
    const bool dense = c->denseAllowed && c->debugMode < 0 && maxLights <= RT_DENSE_MAX_LIGHTS && l.paths2.base != nullptr;
    const bool denseAll = dense && first.lightSamplingStrategy == RT_LIGHT_SAMPLING_ALL && !c->plainPathTracer;
    if (dense)
    {
        const uint32_t shardCapacity = (totalSlots + RT_DENSE_SHARDS - 1u) / RT_DENSE_SHARDS + 65536u;
        const uint32_t plane = 2u * RT_DENSE_SHARDS;
        // a fresh path's records: only origin and direction are stored, bounce 0's shade rebuilds the rest from the slot (rt_dense.inl); RTGPU_FULL_PRIMARY=1: all seven.
        // (the tail kernel never sees bounce 0 -- tailDepthFor returns >= 1 -- and the traversal kernels read origin and direction only)
        static const bool fullPrimaryEnv = getenv("RTGPU_FULL_PRIMARY") && atoi(getenv("RTGPU_FULL_PRIMARY")) != 0;
        const bool leanPrimary = !fullPrimaryEnv;
        HIP_TRY(hipMemsetAsync(l.denseCounts, 0, (size_t)plane * (l.queueCountCapacity + 1u) * sizeof(uint32_t), l.stream));
        {
            LaunchTimer t(c, l.stream, KC_GENERATE);
            hipLaunchKernelGGL(k_generate_dense, grid, block, 0, l.stream, c->sceneDev, passesDev, c->numSlots, l.paths, c->slotPixel, totalSlots, shardCapacity, l.denseCounts, c->counters,
                               leanPrimary ? 0u : 1u);
        }
        const bool haveNee = c->numLights != 0 && !c->plainPathTracer;
        // The fused tail (rt_tail.hip): from bounce `tailDepth` on, one persistent launch takes the batch's remaining paths to their end.  Single-mesh
        // scenes behind the 4-wide walk, one next-event request per vertex.
        const uint32_t tailDepth = tailDepthFor(c, totalSlots, maxRayDepth, denseAll, stackClass);
        for (uint32_t depth = 0; depth <= maxRayDepth + 1u; ++depth)
        {
            const Paths& in = (depth & 1u) ? l.paths2 : l.paths;
            const Paths& out = (depth & 1u) ? l.paths : l.paths2;
            if (tailDepth != 0u && depth == tailDepth)
            {
                const TailArgs args = { l.denseCounts + (size_t)plane * depth, shardCapacity, cursors + depth, c->tune.refillMinIdle, c->tune.otherMinLanes, c->deviceFlags,
                                        (getenv("RTGPU_ANYHIT_FAR_FIRST") && atoi(getenv("RTGPU_ANYHIT_FAR_FIRST")) == 0) ? 0u : 1u };
                static const uint32_t tailBlocksPerCU = getenv("RTGPU_TAIL_BLOCKS_PER_CU") ? (uint32_t)atoi(getenv("RTGPU_TAIL_BLOCKS_PER_CU")) : 4u;   // tuning knob
                uint32_t tailBlocks = (totalSlots + RT_TAIL_PATHS - 1u) / RT_TAIL_PATHS;   // never more blocks than chunks of the whole batch
                if (tailBlocks > c->numCUs * tailBlocksPerCU) tailBlocks = c->numCUs * tailBlocksPerCU;
                const dim3 tailGrid(tailBlocks ? tailBlocks : 1u);
                LaunchTimer t(c, l.stream, KC_TAIL);
#define RT_LAUNCH_TAIL(L, P) hipLaunchKernelGGL((k_tail<L, P>), tailGrid, block, 0, l.stream, c->sceneDev, c->wide, passesDev, c->numSlots, in, args, l.home, c->counters)
                if (c->plainPathTracer) RT_LAUNCH_TAIL(0, true);
                else if (c->leanScene == 1) RT_LAUNCH_TAIL(1, false); else if (c->leanScene == 2) RT_LAUNCH_TAIL(2, false);
                else if (c->leanScene == 3) RT_LAUNCH_TAIL(3, false); else if (c->leanScene == 4) RT_LAUNCH_TAIL(4, false); else RT_LAUNCH_TAIL(0, false);
#undef RT_LAUNCH_TAIL
                break;
            }
            const bool haveClosest = depth <= maxRayDepth, haveShadow = depth > 0 && haveNee;
            if (haveClosest || haveShadow)
            {
                TravTuning tune = c->tune;
                tune.denseCounts = haveClosest ? l.denseCounts + (size_t)plane * depth : nullptr; tune.denseShardCapacity = shardCapacity;
                const uint32_t* tsq = haveShadow ? l.shadowQueues[(depth - 1u) & 1u] : nullptr;
                const uint32_t* tsc = haveShadow ? shadowCounts + (depth - 1u) : nullptr;
#define RT_LAUNCH_TRACE_DENSE(S, C) hipLaunchKernelGGL((k_trace<S, C>), travGrid, block, 0, l.stream, c->sceneDev, in, (const uint32_t*)nullptr, (const uint32_t*)nullptr, tsq, tsc, cursors + depth, c->counters, tune)
                if (useWide(c))
                {
                    // the 4-wide tree serves the launch; what it does not trust goes through the binary-tree kernel right behind it (a small grid: few rays)
                    uint32_t* exactCounts = l.queueCounts + 4 * l.queueCountCapacity;
                    uint32_t* exactShadowCounts = l.queueCounts + 5 * l.queueCountCapacity;
                    uint32_t* exactCursors = l.queueCounts + 6 * l.queueCountCapacity;
                    launchTraceWide(c, l.stream, in, nullptr, nullptr, tsq, tsc, cursors + depth, l.exactQueue, exactCounts + depth, l.exactShadowQueue, exactShadowCounts + depth, 0.0001f,
                                    tune.denseCounts, shardCapacity, true, depth);
                    launchRetrace(c, l, l.stream, in, depth, stackClass, l.queues[0]);
                }
                else
                {
                LaunchTimer t(c, l.stream, KC_TRACE);
                if (stackClass == 24u) { if (c->countIntersections) RT_LAUNCH_TRACE_DENSE(24, true); else RT_LAUNCH_TRACE_DENSE(24, false); }
                else if (stackClass == 32u) { if (c->countIntersections) RT_LAUNCH_TRACE_DENSE(32, true); else RT_LAUNCH_TRACE_DENSE(32, false); }
                else { if (c->countIntersections) RT_LAUNCH_TRACE_DENSE(64, true); else RT_LAUNCH_TRACE_DENSE(64, false); }
                }
#undef RT_LAUNCH_TRACE_DENSE
            }
            // bounce `depth`: shades the live paths; folds the visibility results of the previous bounce's zombies in (the last round does only that)
            const DenseCounts dc = { l.denseCounts + (size_t)plane * depth, l.denseCounts + (size_t)plane * (depth + 1u), shardCapacity, c->deviceFlags,
                                     leanPrimary && depth == 0u ? c->slotPixel : nullptr };
            LaunchTimer t(c, l.stream, KC_SHADE);
#define RT_LAUNCH_SHADE_DENSE(L, P, A) hipLaunchKernelGGL((k_shade_dense<L, P, A>), grid, block, 0, l.stream, c->sceneDev, passesDev, c->numSlots, in, out, dc, \
                                                     l.shadowQueues[depth & 1u], shadowCounts + depth, l.home, c->counters)
            if (c->plainPathTracer) RT_LAUNCH_SHADE_DENSE(0, true, false);
            else if (denseAll) { if (c->leanScene == 1) RT_LAUNCH_SHADE_DENSE(1, false, true); else if (c->leanScene == 2) RT_LAUNCH_SHADE_DENSE(2, false, true); else if (c->leanScene == 4) RT_LAUNCH_SHADE_DENSE(4, false, true); else RT_LAUNCH_SHADE_DENSE(0, false, true); }
            else if (c->leanScene == 1) RT_LAUNCH_SHADE_DENSE(1, false, false); else if (c->leanScene == 2) RT_LAUNCH_SHADE_DENSE(2, false, false);
            else if (c->leanScene == 3) RT_LAUNCH_SHADE_DENSE(3, false, false); else if (c->leanScene == 4) RT_LAUNCH_SHADE_DENSE(4, false, false); else RT_LAUNCH_SHADE_DENSE(0, false, false);
#undef RT_LAUNCH_SHADE_DENSE
        }
        if (c->lastAccumulateLane >= 0 && c->lastAccumulateLane != laneIndex) HIP_TRY(hipStreamWaitEvent(l.stream, c->lanes[c->lastAccumulateLane].accumulated, 0));
        {
            LaunchTimer t(c, l.stream, KC_ACCUMULATE);
            hipLaunchKernelGGL(k_accumulate_home, pixelGrid, block, 0, l.stream, l.home, c->slotPixel, c->numSlots, numPasses, c->sum, c->secondary, c->width, passesDev);
        }
    }
    else
    {
    {
        LaunchTimer t(c, l.stream, KC_GENERATE);
        hipLaunchKernelGGL(k_generate, grid, block, 0, l.stream, c->sceneDev, passesDev, c->numSlots, l.paths, c->slotPixel, totalSlots, l.queues[0], pathCounts + 0, c->counters);
    }
#define RT_LAUNCH_TRACE(S, C) hipLaunchKernelGGL((k_trace<S, C>), travGrid, block, 0, l.stream, c->sceneDev, l.paths, tq, tqc, tsq, tsc, cursors + launchIndex, c->counters, c->tune)
    // bounce k: trace {closest rays of bounce k, NEE rays of bounce k-1} -> shade k; one last trace for the NEE rays of
    // the final bounce
    const uint32_t lastDepth = c->debugMode >= 0 ? 0u : maxRayDepth + 1u;
    for (uint32_t depth = 0; depth <= lastDepth; ++depth)
    {
        const bool haveClosest = depth <= maxRayDepth;
        const bool haveShadow = depth > 0 && c->numLights != 0 && !c->plainPathTracer;
        if (haveClosest || haveShadow)
        {
            const uint32_t* tq = haveClosest ? l.queues[depth & 1u] : nullptr;
            const uint32_t* tqc = haveClosest ? pathCounts + depth : nullptr;
            const uint32_t* tsq = haveShadow ? l.shadowQueues[(depth - 1u) & 1u] : nullptr;
            const uint32_t* tsc = haveShadow ? shadowCounts + (depth - 1u) : nullptr;
            const uint32_t launchIndex = depth;
            if (useWide(c))
            {
                // the re-encoded tree serves the launch; what it does not trust goes through the binary-tree kernel right behind it
                uint32_t* exactCounts = l.queueCounts + 4 * l.queueCountCapacity;
                uint32_t* exactShadowCounts = l.queueCounts + 5 * l.queueCountCapacity;
                uint32_t* exactCursors = l.queueCounts + 6 * l.queueCountCapacity;
                launchTraceWide(c, l.stream, l.paths, tq, tqc, tsq, tsc, cursors + launchIndex, l.exactQueue, exactCounts + launchIndex, l.exactShadowQueue, exactShadowCounts + launchIndex, 0.0001f, nullptr, 0u);
                launchRetrace(c, l, l.stream, l.paths, launchIndex, stackClass, l.queues[(depth + 1u) & 1u]);
            }
            else
            {
                LaunchTimer t(c, l.stream, KC_TRACE);
                if (stackClass == 24u) { if (c->countIntersections) RT_LAUNCH_TRACE(24, true); else RT_LAUNCH_TRACE(24, false); }
                else if (stackClass == 32u) { if (c->countIntersections) RT_LAUNCH_TRACE(32, true); else RT_LAUNCH_TRACE(32, false); }
                else { if (c->countIntersections) RT_LAUNCH_TRACE(64, true); else RT_LAUNCH_TRACE(64, false); }
            }
        }
        if (haveClosest)
        {
            LaunchTimer t(c, l.stream, KC_SHADE);
#define RT_LAUNCH_SHADE(L) hipLaunchKernelGGL((k_shade<L>), grid, block, 0, l.stream, c->sceneDev, passesDev, c->numSlots, l.paths, l.queues[depth & 1u], pathCounts + depth, \
                                             l.queues[(depth + 1u) & 1u], pathCounts + depth + 1, l.shadowQueues[depth & 1u], shadowCounts + depth, c->counters)
            if (c->debugMode >= 0)
                hipLaunchKernelGGL(k_debug_shade, grid, block, 0, l.stream, c->sceneDev, l.paths, l.queues[0], pathCounts + 0, (uint32_t)c->debugMode, c->counters);
            else if (c->plainPathTracer)
            {
                hipLaunchKernelGGL((k_shade<false, true>), grid, block, 0, l.stream, c->sceneDev, passesDev, c->numSlots, l.paths, l.queues[depth & 1u], pathCounts + depth,
                                   l.queues[(depth + 1u) & 1u], pathCounts + depth + 1, l.shadowQueues[depth & 1u], shadowCounts + depth, c->counters);
            }
            else if (c->leanScene == 1) RT_LAUNCH_SHADE(true); else RT_LAUNCH_SHADE(false);
        }
    }
    // the film is summed in pass order: this batch's accumulate runs after the previous batch's
    if (c->lastAccumulateLane >= 0 && c->lastAccumulateLane != laneIndex) HIP_TRY(hipStreamWaitEvent(l.stream, c->lanes[c->lastAccumulateLane].accumulated, 0));
    {
        LaunchTimer t(c, l.stream, KC_ACCUMULATE);
        hipLaunchKernelGGL(k_accumulate, pixelGrid, block, 0, l.stream, l.paths, c->numSlots, numPasses, c->sum, c->secondary, c->width, passesDev, c->counters);
    }
    }
    HIP_TRY(hipEventRecord(l.accumulated, l.stream));
    c->lastAccumulateLane = laneIndex;
    c->pending.erase(c->pending.begin(), c->pending.begin() + numPasses);
    c->batchesSinceSync++;
    // a stream starts with small batches (a caller that renders 4 or 8 passes and reads back gets two or three overlapping launch
    // sequences instead of one: +7 %) and grows while the caller keeps streaming
    if (!c->passBatchFromEnv && c->numSlots >= 400000u && numPasses == c->passBatch && ++c->batchesAtThisSize >= c->numLanes)
    {
        // every lane has one batch of this size in flight: the next round of the lanes carries twice as many passes
        uint32_t next = c->passBatch * 2u;
        if (next > maxStreamingBatch(c)) next = maxStreamingBatch(c);
        if (next > c->passBatch) { c->passBatch = next; c->batchesAtThisSize = 0; }
    }
    HIP_TRY(hipGetLastError());
    for (uint32_t i = 0; i < numPasses; ++i)
    {
        HIP_TRY(hipEventRecord(c->seedEvents[firstSlot + i], l.stream));
        c->seedEventUsed[firstSlot + i] = true;
    }
    return RTGPU_OK;
}

// Submits everything that is queued.  What is left when the caller stops streaming (a synchronising call, a parameter change) goes out
// as one batch per free lane instead of one batch: the launch sequences of the parts overlap, which hides the tails of their persistent
// launches (20 passes between read-backs: 8 + 12 -> 8 + 6 + 6 on three lanes).  Results do not depend on the split.
static int flushPending(RtgpuContext* c)
{
    if (c->pending.empty()) return RTGPU_OK;
    uint32_t parts = 1;
    if (c->numSlots >= 400000u && !c->passBatchFromEnv && !c->vcm.enabled)
    {
        const uint32_t lanesFree = c->batchesSinceSync ? c->numLanes - 1u : c->numLanes;
        parts = (uint32_t)c->pending.size() / 2u;
        if (parts > lanesFree) parts = lanesFree;
        if (parts < 1u) parts = 1u;
    }
    const uint32_t each = ((uint32_t)c->pending.size() + parts - 1u) / parts;
    while (!c->pending.empty()) { const int r = flushBatch(c, each); if (r) return r; }
    return RTGPU_OK;
}

#include "rt_runtime_vcm.inl"   // the bidirectional integrator's and the Light Tracer's launch sequences

static void defaultVcmParams(RtVcmParams& vp)
{
    memset(&vp, 0, sizeof(vp));
    vp.maxPathLength = 10; vp.useVertexConnection = 1; vp.useVertexMerging = 1;
    vp.initialMergingRadius = 0.02f; vp.minMergingRadius = 0.02f; vp.mergingRadiusMultiplier = 1.0f;
    for (int k = 0; k < 4; ++k) vp.bsdfSamplingWeight[k] = vp.lightSamplingWeight[k] = vp.vertexConnectingWeight[k] = vp.cameraConnectingWeight[k] = vp.vertexMergingWeight[k] = 1.0f;
}

RTGPU_API int rtgpu_set_integrator(RtgpuContext* c, uint32_t integrator, const RtVcmParams* vcm)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    if (integrator > RT_INTEGRATOR_LIGHT_TRACER) return fail(RTGPU_ERR_INVALID_ARGUMENT, "unknown integrator");
    if ((!c->peers.empty() || c->isPeer) && (integrator == RT_INTEGRATOR_VCM || integrator == RT_INTEGRATOR_LIGHT_TRACER))
        return fail(RTGPU_ERR_UNSUPPORTED, "VCM and the Light Tracer splat over the whole frame: they need a single-device context (rtgpu_create)");
    RT_FAN_OUT(c, rtgpu_set_integrator(peer, integrator, vcm));
    int r = rtgpu_synchronize(c); if (r) return r;
    RtVcmParams vp; defaultVcmParams(vp);
    if (vcm) vp = *vcm;
    if (integrator == RT_INTEGRATOR_VCM)
    {
        if (vp.maxPathLength < 1u || vp.maxPathLength > RT_VCM_MAX_PATH_LENGTH) return fail(RTGPU_ERR_INVALID_ARGUMENT, "maxPathLength must be 1..16");
        if (!(vp.initialMergingRadius >= vp.minMergingRadius) || !(vp.minMergingRadius > 0.0f)) return fail(RTGPU_ERR_INVALID_ARGUMENT, "merging radii: initial >= min > 0 required");
        if (!(vp.mergingRadiusMultiplier > 0.0f && vp.mergingRadiusMultiplier <= 1.0f)) return fail(RTGPU_ERR_INVALID_ARGUMENT, "mergingRadiusMultiplier must be in (0, 1]");
    }
    c->vcm.enabled = integrator == RT_INTEGRATOR_VCM;
    c->plainPathTracer = integrator == RT_INTEGRATOR_PATH_TRACER;
    c->lightTracer = integrator == RT_INTEGRATOR_LIGHT_TRACER;
    c->debugMode = integrator == RT_INTEGRATOR_DEBUG ? (c->debugMode >= 0 ? c->debugMode : (int)DBG_TRIANGLE_ID) : -1;   // DebugRenderer's default mode, DebugRenderer.cpp:16
    c->vcm.params = vp;
    c->vcm.havePhotons = false;
    return RTGPU_OK;
}

RTGPU_API int rtgpu_set_debug_rendering_mode(RtgpuContext* c, uint32_t mode)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    if (mode >= DBG_NUM_MODES) return fail(RTGPU_ERR_INVALID_ARGUMENT, "unknown DebugRenderingMode");
    if (c->debugMode < 0) return fail(RTGPU_ERR_NOT_READY, "the integrator is not RT_INTEGRATOR_DEBUG");
    RT_FAN_OUT(c, rtgpu_set_debug_rendering_mode(peer, mode));
    int r = rtgpu_synchronize(c); if (r) return r;
    c->debugMode = (int)mode;
    return RTGPU_OK;
}

RTGPU_API int rtgpu_vcm_num_photons(RtgpuContext* c, uint32_t* outCount)
{
    if (!c || !outCount) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    int r = rtgpu_synchronize(c); if (r) return r;
    *outCount = 0;
    if (!c->vcm.havePhotons || !c->vcm.arena.photonCount) return RTGPU_OK;
    std::vector<uint32_t> counts(c->numSlots);
    HIP_TRY(rtMemcpy(counts.data(), c->vcm.arena.photonCount + (size_t)c->vcm.lastPhotonPass * c->numSlots, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    unsigned long long total = 0; for (uint32_t n : counts) total += n;
    *outCount = (uint32_t)total;
    return RTGPU_OK;
}

RTGPU_API int rtgpu_render_pass(RtgpuContext* c, const RtPassParams* p)
{
    if (!c || !p) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->sceneReady) return fail(RTGPU_ERR_NOT_READY, "rtgpu_upload_scene has not been called");
    if (!c->sum) return fail(RTGPU_ERR_NOT_READY, "rtgpu_resize has not been called");
    if (p->numDimensions > RTGPU_MAX_DIMENSIONS) return fail(RTGPU_ERR_INVALID_ARGUMENT, "numDimensions exceeds RTGPU_MAX_DIMENSIONS");
    if (p->numDimensions > 0 && !p->seed) return fail(RTGPU_ERR_INVALID_ARGUMENT, "seed is NULL");
    if (p->maxRayDepth >= 255u) return fail(RTGPU_ERR_INVALID_ARGUMENT, "maxRayDepth must be < 255");
    if (p->camera.dofEnable && p->camera.bokehShape > 2u) return fail(RTGPU_ERR_UNSUPPORTED, "bokeh shapes: circle, hexagon, square (NGon is a TODO in the reference, texture-shaped bokeh is not implemented)");
    RT_FAN_OUT(c, rtgpu_render_pass(peer, p));   // asynchronous on every device: the shards render side by side
    HIP_TRY(hipSetDevice(c->device));
    if (c->numSlots == 0) return RTGPU_OK;   // this shard owns no pixels
    if (c->vcm.enabled) return vcmRenderPass(c, p);
    if (c->lightTracer) return lightTracerRenderPass(c, p);

    CtxPending pd;
    DevPass& pass = pd.pass;
    memset(&pass, 0, sizeof(pass));
    pass.camera = p->camera;
    pass.seed = nullptr;   // assigned when the batch is submitted
    pass.numDimensions = p->numDimensions;
    pass.blueNoiseLayers = (c->sceneDev.blueNoise && p->useBlueNoise) ? 4u : 0u;   // GenericSampler.cpp:69-73
    pass.sampleOffset[0] = p->sampleOffset[0]; pass.sampleOffset[1] = p->sampleOffset[1];
    pass.passIndex = p->passIndex;
    pass.maxRayDepth = p->maxRayDepth;
    pass.minRussianRouletteDepth = p->minRussianRouletteDepth;
    pass.lightSamplingStrategy = p->lightSamplingStrategy;
    memcpy(pass.lightSamplingWeight, p->lightSamplingWeight, 16);
    memcpy(pass.bsdfSamplingWeight, p->bsdfSamplingWeight, 16);
    pass.rngKey[0] = p->rngKey[0]; pass.rngKey[1] = p->rngKey[1];
    pass.width = c->width; pass.height = c->height;
    pd.seeds.assign(p->seed, p->seed + p->numDimensions);   // the caller's array may be reused right away

    // all passes of a batch share the structural parameters; a change submits what is queued first
    if (!c->pending.empty())
    {
        const DevPass& f = c->pending[0].pass;
        const bool same = f.numDimensions == pass.numDimensions && f.blueNoiseLayers == pass.blueNoiseLayers && f.maxRayDepth == pass.maxRayDepth &&
                          f.minRussianRouletteDepth == pass.minRussianRouletteDepth && f.lightSamplingStrategy == pass.lightSamplingStrategy &&
                          memcmp(f.lightSamplingWeight, pass.lightSamplingWeight, 16) == 0 && memcmp(f.bsdfSamplingWeight, pass.bsdfSamplingWeight, 16) == 0;
        if (!same) { int r = flushPending(c); if (r) return r; }
    }
    c->pending.push_back(std::move(pd));
    const uint32_t lightLimit = maxBatchFor(c, pass.lightSamplingStrategy == RT_LIGHT_SAMPLING_ALL ? c->numLights : 1u);
    if (c->pending.size() >= (c->passBatch < lightLimit ? c->passBatch : lightLimit)) return flushBatch(c, 0u);
    return RTGPU_OK;
}

RTGPU_API int rtgpu_synchronize(RtgpuContext* c)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    // every device's queued passes -- the first device's included -- are submitted before the first wait, so that the tails run side by side
    for (RtgpuContext* peer : c->peers) { HIP_TRY(hipSetDevice(peer->device)); int r = flushPending(peer); if (r) return r; }
    HIP_TRY(hipSetDevice(c->device));
    { int r = vcmFlush(c); if (r) return r; }
    { int r = flushPending(c); if (r) return r; }
    RT_FAN_OUT(c, rtgpu_synchronize(peer));
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(syncLanes(c));
    c->batchesSinceSync = 0; c->batchesAtThisSize = 0;
    if (!c->passBatchFromEnv) c->passBatch = c->passBatchBase;
    if (c->deviceFlags && c->deviceFlags[0] != 0u)
    {
        c->deviceFlags[0] = 0u;
        return fail(RTGPU_ERR_DEVICE, "dense path state: a region of the arena overflowed (the frame since the last reset is invalid)");
    }
    return resolveTimed(c);
}

RTGPU_API int rtgpu_read_sum(RtgpuContext* c, float* sumRGB, float* secondaryRGB)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    if (!c->sum) return fail(RTGPU_ERR_NOT_READY, "rtgpu_resize has not been called");
    int r = rtgpu_synchronize(c); if (r) return r;
    r = gatherPeers(c); if (r) return r;
    const size_t bytes = (size_t)c->width * c->height * 3 * sizeof(float);
    if (sumRGB) HIP_TRY(rtMemcpy(sumRGB, c->sum, bytes, hipMemcpyDeviceToHost));
    if (secondaryRGB) HIP_TRY(rtMemcpy(secondaryRGB, c->secondary, bytes, hipMemcpyDeviceToHost));
    return RTGPU_OK;
}

// page-locked host memory for the read-back calls: hipMemcpy into registered memory runs at the PCIe link's rate
RTGPU_API int rtgpu_host_register(RtgpuContext* c, void* ptr, size_t bytes)
{
    if (!c || !ptr || !bytes) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(e == hipErrorNoDevice || e == hipErrorInvalidDevice ? RTGPU_ERR_NO_DEVICE : RTGPU_ERR_DEVICE, std::string("hipHostRegister: ") + hipGetErrorString(e)); }
    return RTGPU_OK;
}

RTGPU_API int rtgpu_host_unregister(RtgpuContext* c, void* ptr)
{
    if (!c || !ptr) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(RTGPU_ERR_DEVICE, std::string("hipHostUnregister: ") + hipGetErrorString(e)); }
    return RTGPU_OK;
}

RTGPU_API int rtgpu_get_device_sum(RtgpuContext* c, void** sumDevice, void** secondaryDevice, size_t* numFloats)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    if (!c->sum) return fail(RTGPU_ERR_NOT_READY, "rtgpu_resize has not been called");
    int r = rtgpu_synchronize(c); if (r) return r;   // the pointers are handed out with every queued pass accumulated
    r = gatherPeers(c); if (r) return r;
    if (sumDevice) *sumDevice = c->sum;
    if (secondaryDevice) *secondaryDevice = c->secondary;
    if (numFloats) *numFloats = (size_t)c->width * c->height * 3;
    return RTGPU_OK;
}

RTGPU_API int rtgpu_get_counters(RtgpuContext* c, RtCounters* out)
{
    if (!c || !out) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    int r = rtgpu_synchronize(c); if (r) return r;
    unsigned long long host[16];
    HIP_TRY(rtMemcpy(host, c->counters, sizeof(host), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    out->numRays = host[C_RAYS]; out->numShadowRays = host[C_SHADOW]; out->numShadowRaysHit = host[C_SHADOW_HIT];
    out->numPrimaryRays = host[C_PRIMARY]; out->numRayBoxTests = host[C_BOX]; out->numPassedRayBoxTests = host[C_BOX_PASS];
    out->numRayTriangleTests = host[C_TRI]; out->numPassedRayTriangleTests = host[C_TRI_PASS];
    out->numMeshHits = host[C_MESH_HITS]; out->numAnalyticHits = host[C_ANALYTIC_HITS];
    out->numShadowRayBoxTests = host[C_BOX_SHADOW]; out->numShadowRayTriangleTests = host[C_TRI_SHADOW];
    out->numRetracedRays = host[RT_COUNTER_RETRACED];
    out->_reserved[0] = host[RT_COUNTER_RETRACED + 1]; out->_reserved[1] = host[RT_COUNTER_RETRACED + 2]; out->_reserved[2] = host[RT_COUNTER_RETRACED + 3];   // diagnostics of the re-encoded walks: untrusted rays, stack overflows
    for (RtgpuContext* peer : c->peers)
    {
        RtCounters pc;
        r = rtgpu_get_counters(peer, &pc); if (r) return r;
        uint64_t* a = (uint64_t*)out; const uint64_t* b = (const uint64_t*)&pc;
        for (size_t i = 0; i < sizeof(RtCounters) / sizeof(uint64_t); ++i) a[i] += b[i];
    }
    HIP_TRY(hipSetDevice(c->device));
    return RTGPU_OK;
}

RTGPU_API int rtgpu_set_intersection_counters(RtgpuContext* c, int enable)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    RT_FAN_OUT(c, rtgpu_set_intersection_counters(peer, enable));
    int r = rtgpu_synchronize(c); if (r) return r;
    c->countIntersections = enable != 0;
    return RTGPU_OK;
}

static int checkBlocks(RtgpuContext* c, uint32_t numBlocks, const RtBlock* blocks)
{
    if (numBlocks && !blocks) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    for (uint32_t i = 0; i < numBlocks; ++i)
        if (blocks[i].minX >= blocks[i].maxX || blocks[i].minY >= blocks[i].maxY || blocks[i].maxX > c->width || blocks[i].maxY > c->height)
            return fail(RTGPU_ERR_INVALID_ARGUMENT, "block outside the viewport or empty");
    return RTGPU_OK;
}

RTGPU_API int rtgpu_compute_block_errors(RtgpuContext* c, uint32_t numPasses, uint32_t numBlocks, const RtBlock* blocks, float* outErrors)
{
    if (!c || (numBlocks && !outErrors)) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->sum) return fail(RTGPU_ERR_NOT_READY, "rtgpu_resize has not been called");
    if (numPasses == 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "numPasses must be > 0");
    int r = checkBlocks(c, numBlocks, blocks); if (r) return r;
    if (numBlocks == 0) return RTGPU_OK;
    r = rtgpu_synchronize(c); if (r) return r;
    r = gatherPeers(c); if (r) return r;
    std::vector<ErrorRow> rows; std::vector<uint32_t> firstRow(numBlocks);
    for (uint32_t i = 0; i < numBlocks; ++i)
    {
        firstRow[i] = (uint32_t)rows.size();
        for (uint32_t y = blocks[i].minY; y < blocks[i].maxY; ++y) rows.push_back({ i, y });
    }
    RtBlock* dBlocks = nullptr; ErrorRow* dRows = nullptr; uint32_t* dFirst = nullptr; float* dRowErrors = nullptr; float* dOut = nullptr;
    hipError_t e = hipMalloc((void**)&dBlocks, numBlocks * sizeof(RtBlock));
    if (e == hipSuccess) e = hipMalloc((void**)&dRows, rows.size() * sizeof(ErrorRow));
    if (e == hipSuccess) e = hipMalloc((void**)&dFirst, numBlocks * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&dRowErrors, rows.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&dOut, numBlocks * sizeof(float));
    if (e == hipSuccess) e = rtMemcpy(dBlocks, blocks, numBlocks * sizeof(RtBlock), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rtMemcpy(dRows, rows.data(), rows.size() * sizeof(ErrorRow), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rtMemcpy(dFirst, firstRow.data(), numBlocks * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess)
    {
        hipStream_t st = c->lanes[0].stream;
        const uint32_t numRows = (uint32_t)rows.size();
        hipLaunchKernelGGL(k_block_error_rows, dim3((numRows + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, st, c->sum, c->secondary, c->width, dBlocks, dRows, numRows,
                           1.0f / (float)numPasses, dRowErrors);
        hipLaunchKernelGGL(k_block_error_total, dim3((numBlocks + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, st, dBlocks, dFirst, numBlocks, dRowErrors, c->width * c->height, dOut);
        e = hipStreamSynchronize(st);
    }
    if (e == hipSuccess) e = rtMemcpy(outErrors, dOut, numBlocks * sizeof(float), hipMemcpyDeviceToHost);
    for (void* p : { (void*)dBlocks, (void*)dRows, (void*)dFirst, (void*)dRowErrors, (void*)dOut }) if (p) (void)hipFree(p);
    if (e != hipSuccess) return fail(RTGPU_ERR_DEVICE, std::string("rtgpu_compute_block_errors: ") + hipGetErrorString(e));
    return RTGPU_OK;
}

RTGPU_API int rtgpu_set_active_blocks(RtgpuContext* c, uint32_t numBlocks, const RtBlock* blocks)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    if (!c->sum) return fail(RTGPU_ERR_NOT_READY, "rtgpu_resize has not been called");
    int r = checkBlocks(c, numBlocks, blocks); if (r) return r;
    RT_FAN_OUT(c, rtgpu_set_active_blocks(peer, numBlocks, blocks));
    r = rtgpu_synchronize(c); if (r) return r;
    c->activeMask.clear();
    if (numBlocks)
    {
        c->activeMask.assign((size_t)c->width * c->height, 0);
        for (uint32_t i = 0; i < numBlocks; ++i)
            for (uint32_t y = blocks[i].minY; y < blocks[i].maxY; ++y)
                for (uint32_t x = blocks[i].minX; x < blocks[i].maxX; ++x)
                {
                    uint8_t& m = c->activeMask[(size_t)y * c->width + x];
                    if (m) return fail(RTGPU_ERR_INVALID_ARGUMENT, "active blocks overlap");
                    m = 1;
                }
    }
    return rebuildSlots(c);
}

RTGPU_API int rtgpu_postprocess(RtgpuContext* c, const RtPostprocessParams* p, uint32_t* frontBufferBGRA)
{
    if (!c || !p || !frontBufferBGRA) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->sum) return fail(RTGPU_ERR_NOT_READY, "rtgpu_resize has not been called");
    // bloom: the reference's blur works on 4 columns at a time and on 4096-entry line buffers without bounds checks
    // (Bitmap.cpp:925-931, :941, :978-990): sizes it would read or write out of bounds for are refused here
    const bool bloom = p->bloomFactor > 0.0f;
    BlurPlan plans[5];
    if (bloom)
    {
        if (c->width > 4096u || c->height > 4096u || (c->width % 4u) != 0u) return fail(RTGPU_ERR_UNSUPPORTED, "bloom: width must be a multiple of 4 and both sizes <= 4096 (Bitmap::GaussianBlur)");
        float blurSigma = 2.0f;   // Viewport.cpp:438-444
        for (int l = 0; l < 5; ++l)
        {
            const uint32_t n = 8u;
            const float sigma = blurSigma;
            float wIdeal = sqrtf((12.0f * sigma * sigma / n) + 1.0f);   // Bitmap.cpp:935-946
            uint32_t wl = (uint32_t)floorf(wIdeal);
            if (wl % 2u == 0u) wl--;
            const uint32_t wu = wl + 2u;
            const float mIdeal = (12.0f * sigma * sigma - n * wl * wl - 4.0f * n * wl - 3.0f * n) / (-4.0f * wl - 4.0f);
            plans[l].n = n; plans[l].wl = wl; plans[l].wu = wu; plans[l].m = roundf(mIdeal);
            if (c->width <= 2u * wu + 1u || c->height <= 2u * wu + 1u) return fail(RTGPU_ERR_UNSUPPORTED, "bloom: the image is smaller than the widest blur window (2 * 97 + 1 pixels)");
            blurSigma *= 2.5f;
        }
    }
    if (p->tonemapper > RT_TONEMAPPER_ACES) return fail(RTGPU_ERR_INVALID_ARGUMENT, "unknown tonemapper");
    if (p->numPasses == 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "numPasses must be > 0");
    int r = rtgpu_synchronize(c); if (r) return r;
    r = gatherPeers(c); if (r) return r;
    const size_t pixels = (size_t)c->width * c->height;
    uint32_t* dFront = nullptr;
    HIP_TRY(hipMalloc((void**)&dFront, pixels * sizeof(uint32_t)));
    const float exposureScale = powf(2.0f, p->exposure);   // colorScale on the host like the reference (Viewport.cpp:453)
    const PostScale scale = { { p->colorFilter[0] * exposureScale, p->colorFilter[1] * exposureScale, p->colorFilter[2] * exposureScale } };
    hipStream_t stream = c->lanes[0].stream;
    hipError_t e = hipSuccess;
    float* dBlur = nullptr; float* dLines = nullptr;
    if (!bloom) hipLaunchKernelGGL(k_postprocess, dim3((uint32_t)((pixels + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, stream, c->sum, dFront, c->width, c->height, *p, scale);
    else
    {
        // mBlurredImages[i] = GaussianBlur(copy of (i == 0 ? mSum : mBlurredImages[i - 1]), sigma_i, 8), Viewport.cpp:436-445
        e = hipMalloc((void**)&dBlur, pixels * 3 * sizeof(float) * 5);
        if (e == hipSuccess) e = hipMalloc((void**)&dLines, pixels * 3 * sizeof(float) * 2);
        BloomLevels levels;
        for (int l = 0; l < 5 && e == hipSuccess; ++l)
        {
            float* img = dBlur + (size_t)l * pixels * 3;
            levels.level[l] = img;
            e = hipMemcpyAsync(img, l == 0 ? c->sum : dBlur + (size_t)(l - 1) * pixels * 3, pixels * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(k_blur_lines, dim3((c->height * 3u + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, stream, img, c->width, c->height, 0u, plans[l], dLines, dLines + pixels * 3);
            hipLaunchKernelGGL(k_blur_lines, dim3((c->width * 3u + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, stream, img, c->width, c->height, 1u, plans[l], dLines, dLines + pixels * 3);
        }
        if (e == hipSuccess) hipLaunchKernelGGL(k_postprocess_bloom, dim3((uint32_t)((pixels + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, stream, c->sum, levels, dFront, c->width, c->height, *p, scale);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = rtMemcpy(frontBufferBGRA, dFront, pixels * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(dFront);
    if (dBlur) (void)hipFree(dBlur);
    if (dLines) (void)hipFree(dLines);
    if (e != hipSuccess) return fail(RTGPU_ERR_DEVICE, std::string("rtgpu_postprocess: ") + hipGetErrorString(e));
    return RTGPU_OK;
}

RTGPU_API int rtgpu_evaluate_textures(RtgpuContext* c, uint32_t count, const uint32_t* textureIndex, const float* uv, float* out)
{
    if (!c || (count && (!textureIndex || !uv || !out))) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->sceneReady) return fail(RTGPU_ERR_NOT_READY, "rtgpu_upload_scene has not been called");
    if (count == 0) return RTGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    for (uint32_t i = 0; i < count; ++i) if (textureIndex[i] >= c->sceneDev.numTextures) return fail(RTGPU_ERR_INVALID_ARGUMENT, "texture index out of range");
    uint32_t* dIndex = nullptr; float* dUv = nullptr; float* dOut = nullptr;
    hipError_t e = hipMalloc((void**)&dIndex, count * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&dUv, (size_t)count * 2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&dOut, (size_t)count * 4 * sizeof(float));
    if (e == hipSuccess) e = rtMemcpy(dIndex, textureIndex, count * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rtMemcpy(dUv, uv, (size_t)count * 2 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(k_evaluate_textures, dim3((count + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, c->lanes[0].stream, c->sceneDev, count, dIndex, dUv, dOut);
        e = hipStreamSynchronize(c->lanes[0].stream);
    }
    if (e == hipSuccess) e = rtMemcpy(out, dOut, (size_t)count * 4 * sizeof(float), hipMemcpyDeviceToHost);
    if (dIndex) (void)hipFree(dIndex);
    if (dUv) (void)hipFree(dUv);
    if (dOut) (void)hipFree(dOut);
    if (e != hipSuccess) return fail(RTGPU_ERR_DEVICE, std::string("rtgpu_evaluate_textures: ") + hipGetErrorString(e));
    return RTGPU_OK;
}

RTGPU_API int rtgpu_kat(RtgpuContext* c, uint32_t func, const float* in, uint32_t inStride, float* out, uint32_t outStride, uint32_t n)
{
    if (!c || (n && (!in || !out))) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n == 0) return RTGPU_OK;
    static const struct { uint32_t func, minIn, minOut; } known[] = {
        { KAT_SIN_LANE, 1, 1 }, { KAT_SINCOS, 1, 4 }, { KAT_FASTLOG, 1, 1 }, { KAT_FASTACOS, 1, 1 }, { KAT_FASTATAN2, 2, 1 }, { KAT_FLOAT_NORMAL2, 2, 4 },
        { KAT_HEMISPHERE_COS, 2, 4 }, { KAT_SPHERE, 2, 4 }, { KAT_CIRCLE, 2, 4 }, { KAT_ORTHO_BASIS, 4, 8 }, { KAT_FRESNEL_DIELECTRIC, 2, 1 },
        { KAT_FRESNEL_METAL, 3, 1 }, { KAT_REFRACT3, 9, 4 }, { KAT_REFLECT3, 8, 4 }, { KAT_BOX_RAY, 14, 2 }, { KAT_BOX_RAY_TWOSIDED, 14, 3 },
        { KAT_TRIANGLE_RAY, 17, 4 }, { KAT_MAKE_RAY, 8, 12 }, { KAT_TRANSFORM_RAY, 24, 16 }, { KAT_FAST_INVERSE, 16, 16 }, { KAT_TRANSFORM_SCALED, 20, 12 }, { KAT_FRAME_COMPOSE, 40, 20 }, { KAT_SHAPE_INTERSECT, 13, 4 },
        { KAT_SHAPE_SAMPLE, 12, 8 }, { KAT_SHAPE_PDF, 13, 1 }, { KAT_SHAPE_EVAL, 13, 16 },
        { KAT_LIGHT_ILLUMINATE, (uint32_t)(sizeof(RtLight) / 4) + 19, 11 }, { KAT_LIGHT_RADIANCE, (uint32_t)(sizeof(RtLight) / 4) + 13, 5 },
        { KAT_LIGHT_EMIT, (uint32_t)(sizeof(RtLight) / 4) + 5, 15 }, { KAT_LIGHT_ILLUMINATE_BIDIR, (uint32_t)(sizeof(RtLight) / 4) + 19, 12 },
        { KAT_LIGHT_RADIANCE_BIDIR, (uint32_t)(sizeof(RtLight) / 4) + 13, 6 }, { KAT_BSDF_SAMPLE, 23, 11 }, { KAT_BSDF_EVALUATE, 24, 5 }, { KAT_BSDF_PDFS, 24, 8 },
        { KAT_CAMERA_RAY, (uint32_t)(sizeof(RtCamera) / 4) + 8, 16 }, { KAT_CAMERA_FILM, (uint32_t)(sizeof(RtCamera) / 4) + 8, 6 }, { KAT_FILM_SPLAT, 12, 10 },
        { KAT_PACKED_PHOTON, 8, 11 }, { KAT_HSV_TO_RGB, 2, 4 } };
    bool ok = false;
    for (const auto& k : known) if (k.func == func) { if (inStride < k.minIn || outStride < k.minOut) return fail(RTGPU_ERR_INVALID_ARGUMENT, "rtgpu_kat: record stride too small for this function"); ok = true; }
    if (!ok) return fail(RTGPU_ERR_INVALID_ARGUMENT, "rtgpu_kat: unknown function id");
    RtSceneDesc none; memset(&none, 0, sizeof(none));   // the fixtures' lights and materials carry no textures
    return katRoundTrip(c, in, (size_t)n * inStride * 4, out, (size_t)n * outStride * 4, [&](void* dIn, void* dOut, hipStream_t st) {
        hipLaunchKernelGGL(k_kat, dim3((n + 63u) / 64u), dim3(64), 0, st, none, func, (const float*)dIn, inStride, (float*)dOut, outStride, n);
    });
}

RTGPU_API int rtgpu_kat_sampler(RtgpuContext* c, const uint16_t* blueNoise, const uint32_t* in, uint32_t inStride, uint32_t count, uint32_t n, uint32_t* outInts, float* outFloats)
{
    if (!c || !in || !outInts || !outFloats || n == 0 || count == 0) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    for (uint32_t r = 0; r < n; ++r) if (inStride < 4u + in[(size_t)r * inStride + 3]) return fail(RTGPU_ERR_INVALID_ARGUMENT, "rtgpu_kat_sampler: record shorter than its seed table");
    HIP_TRY(hipSetDevice(c->device));
    uint16_t* dBlue = nullptr;
    if (blueNoise)
    {
        HIP_TRY(hipMalloc((void**)&dBlue, (size_t)128 * 128 * 4 * sizeof(uint16_t)));
        const hipError_t e = rtMemcpy(dBlue, blueNoise, (size_t)128 * 128 * 4 * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(dBlue); return fail(RTGPU_ERR_DEVICE, hipGetErrorString(e)); }
    }
    std::vector<float> out((size_t)n * 2 * count);
    const int r = katRoundTrip(c, in, (size_t)n * inStride * 4, out.data(), out.size() * 4, [&](void* dIn, void* dOut, hipStream_t st) {
        hipLaunchKernelGGL(k_kat_sampler, dim3((n + 63u) / 64u), dim3(64), 0, st, dBlue, (const float*)dIn, inStride, (float*)dOut, count, n);
    });
    if (dBlue) (void)hipFree(dBlue);
    if (r) return r;
    for (uint32_t k = 0; k < n; ++k)
    {
        memcpy(outInts + (size_t)k * count, out.data() + (size_t)k * 2 * count, count * 4);
        memcpy(outFloats + (size_t)k * count, out.data() + (size_t)k * 2 * count + count, count * 4);
    }
    return RTGPU_OK;
}

RTGPU_API int rtgpu_kat_mesh(RtgpuContext* c, const float* rays, uint32_t n, uint32_t* out)
{
    if (!c || (n && (!rays || !out))) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->sceneReady) return fail(RTGPU_ERR_NOT_READY, "rtgpu_upload_scene has not been called");
    if (c->sceneDev.numObjects != 1u || c->sceneDev.numMeshes != 1u) return fail(RTGPU_ERR_INVALID_ARGUMENT, "rtgpu_kat_mesh needs a scene made of exactly one mesh object");
    if (c->traversalStackNeed > RT_KAT_MESH_STACK) return fail(RTGPU_ERR_UNSUPPORTED, "mesh BVH deeper than the KAT kernel's stack");
    if (n == 0) return RTGPU_OK;
    { int fr = flushPending(c); if (fr) return fr; }
    return katRoundTrip(c, rays, (size_t)n * 7 * 4, out, (size_t)n * 19 * 4, [&](void* dIn, void* dOut, hipStream_t st) {
        hipLaunchKernelGGL(k_kat_mesh, dim3((n + 63u) / 64u), dim3(64), 0, st, c->sceneDev, (const float*)dIn, n, (uint32_t*)dOut);
    });
}

RTGPU_API int rtgpu_set_concurrency(RtgpuContext* c, uint32_t lanes)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    if (lanes < 1 || lanes > RT_MAX_LANES) return fail(RTGPU_ERR_INVALID_ARGUMENT, "lanes must be 1..6");
    RT_FAN_OUT(c, rtgpu_set_concurrency(peer, lanes));
    int r = rtgpu_synchronize(c); if (r) return r;
    c->numLanes = lanes; c->nextLane = 0; c->lanesChosen = true;
    return RTGPU_OK;
}

RTGPU_API int rtgpu_set_schedule(RtgpuContext* c, uint32_t knob, int32_t value)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    if (knob > RTGPU_SCHEDULE_LOCAL_RETRACE || value < -1 || (knob == RTGPU_SCHEDULE_LOCAL_RETRACE && value > 1) || value > 254) return fail(RTGPU_ERR_INVALID_ARGUMENT, "unknown knob or value out of range");
    RT_FAN_OUT(c, rtgpu_set_schedule(peer, knob, value));
    int r = rtgpu_synchronize(c); if (r) return r;
    if (knob == RTGPU_SCHEDULE_TAIL_BOUNCE) c->tailBounce = value; else c->localRetrace = value;
    return RTGPU_OK;
}

RTGPU_API int rtgpu_enable_timing(RtgpuContext* c, int enable)
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    RT_FAN_OUT(c, rtgpu_enable_timing(peer, enable));
    int r = rtgpu_synchronize(c); if (r) return r;
    c->timing = enable != 0;
    return RTGPU_OK;
}

RTGPU_API int rtgpu_get_walk_info(RtgpuContext* c, RtWalkInfo* out)
{
    if (!c || !out) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL argument");
    const uint32_t kernel = !useWide(c) ? RTGPU_WALK_BINARY : (c->wide.nodes ? RTGPU_WALK_WIDE : RTGPU_WALK_WIDE2);
    out->kernel = kernel; out->reserved = 0u;
    out->nodeBytes = c->walkNodeBytes[kernel]; out->leafBoxBytes = c->walkLeafBoxBytes[kernel]; out->triangleBytes = c->walkTriangleBytes;
    return RTGPU_OK;
}

RTGPU_API int rtgpu_get_kernel_times(RtgpuContext* c, double ms[RTGPU_NUM_KERNEL_CLASSES], uint64_t launches[RTGPU_NUM_KERNEL_CLASSES],
                                     const char* names[RTGPU_NUM_KERNEL_CLASSES])
{
    if (!c) return fail(RTGPU_ERR_INVALID_ARGUMENT, "NULL context");
    int r = rtgpu_synchronize(c); if (r) return r;
    for (int i = 0; i < RTGPU_NUM_KERNEL_CLASSES; ++i)
    {
        if (ms) ms[i] = c->kernelMs[i];
        if (launches) launches[i] = c->kernelLaunches[i];
        if (names) names[i] = kKernelClassNames[i];
    }
    return RTGPU_OK;
}

} // extern "C"
