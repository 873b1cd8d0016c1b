// rt_runtime_vcm.inl -- host side of the bidirectional integrator (renderer "VCM") and of the Light Tracer: arenas, the per-batch launch sequence of
// rt_vcm.inl's kernels, PreRender.  Included by rt_runtime.hip (inside its extern "C" block, after the PathTracerMIS launch sequence).
// =====================================================================================================
// Bidirectional integrator: host side (kernels in rt_vcm.inl)
// =====================================================================================================
#define RT_VCM_COUNT_PLANE (RT_VCM_MAX_PATH_LENGTH + 4u)
#define RT_VCM_NUM_COUNT_PLANES 16u   // 0-9 as before; 10-11 / 12-13 / 14-15: per trace launch the hand-over counts (closest, any-hit) and cursor of the 4-wide walks

// the kernels of rt_vcm.inl that may evaluate textures exist per scene class (rt_shade_kernels.h): a scene without textures takes class 3
// (RTGPU_VCM_CLASS=0: the generic kernels for every scene)
#define RT_LAUNCH_VCM(K, ...) { if (vcmUntextured(c)) hipLaunchKernelGGL((K<3>), __VA_ARGS__); else hipLaunchKernelGGL((K<0>), __VA_ARGS__); }
static bool vcmUntextured(const RtgpuContext* c)
{
    static const bool allow = !(getenv("RTGPU_VCM_CLASS") && atoi(getenv("RTGPU_VCM_CLASS")) == 0);
    return allow && (c->leanScene == 1 || c->leanScene == 3);
}
static void freeVcm(RtgpuContext* c)
{
    RtgpuContext::Vcm& v = c->vcm;
    void* ptrs[] = { v.lightPaths.base, v.cameraPaths.base, v.arena.recs, v.arena.lightVertices, v.arena.photonRaw, v.arena.lvCount, v.arena.photonCount, v.arena.cameraVertex, v.mergeQueue, v.connectQueue, v.overflowQueue, v.exactQueue, v.exactShadowQueue,
                     v.queues[0], v.queues[1], v.queues[2], v.queues[3], v.shadowQueues[0], v.shadowQueues[1], v.shadowQueues[2], v.shadowQueues[3], v.counts,
                     v.passDev, v.seedDev, v.devsDev, v.gridsDev };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (VcmPhotonGrid& g : v.grids) vcmFreePhotonGrid(g);
    const bool enabled = v.enabled; const RtVcmParams params = v.params; const uint32_t batch = v.batch;
    std::vector<RtgpuContext::Vcm::Pending> pending; pending.swap(v.pending);
    v = RtgpuContext::Vcm();
    v.enabled = enabled; v.params = params; v.batch = batch; v.pending.swap(pending);
}

// arenas for `batch` passes of numSlots pixels each
static int ensureVcm(RtgpuContext* c, uint32_t maxLV, uint32_t batch)
{
    RtgpuContext::Vcm& v = c->vcm;
    const uint32_t requests = c->numLights + maxLV;
    const size_t cap = (size_t)(c->numSlots ? c->numSlots : 1) * batch;
    if (v.arena.recs && v.arena.capacity >= cap && v.arena.maxLV >= maxLV && v.requestsPerVertex >= requests && v.batchCapacity >= batch) return RTGPU_OK;
    HIP_TRY(syncLanes(c));
    freeVcm(c);
    if ((unsigned long long)cap * (requests ? requests : 1u) >= 0xFFFFFFFFull) return fail(RTGPU_ERR_UNSUPPORTED, "pixels x passes x shadow requests per vertex exceeds the request index range");
    HIP_TRY(hipMalloc((void**)&v.lightPaths.base, ((size_t)R_NUM_BASE + RT_SHADOW_RECORDS) * cap * sizeof(float4)));
    v.lightPaths.capacity = (uint32_t)cap; v.lightPaths.maxLights = 1;
    HIP_TRY(hipMalloc((void**)&v.cameraPaths.base, ((size_t)R_NUM_BASE + (size_t)(requests ? requests : 1u) * RT_SHADOW_RECORDS) * cap * sizeof(float4)));
    v.cameraPaths.capacity = (uint32_t)cap; v.cameraPaths.maxLights = requests ? requests : 1u;
    HIP_TRY(hipMalloc((void**)&v.arena.recs, (size_t)V_NUM * cap * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&v.arena.lightVertices, (size_t)maxLV * RT_VCM_LV_RECORDS * cap * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&v.arena.photonRaw, (size_t)maxLV * 2 * cap * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&v.arena.cameraVertex, (size_t)RT_VCM_LV_RECORDS * cap * sizeof(float4)));
    HIP_TRY(hipMalloc((void**)&v.mergeQueue, cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&v.connectQueue, cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&v.overflowQueue, cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&v.exactQueue, cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&v.exactShadowQueue, cap * (size_t)(requests ? requests : 1u) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&v.arena.lvCount, cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&v.arena.photonCount, cap * sizeof(uint32_t)));
    HIP_TRY(hipMemset(v.arena.photonCount, 0, cap * sizeof(uint32_t)));
    HIP_TRY(hipStreamSynchronize(nullptr));
    v.arena.capacity = (uint32_t)cap; v.arena.maxLV = maxLV;
    for (int k = 0; k < 4; ++k) HIP_TRY(hipMalloc((void**)&v.queues[k], cap * sizeof(uint32_t)));
    for (int k = 0; k < 2; ++k) HIP_TRY(hipMalloc((void**)&v.shadowQueues[k], cap * sizeof(uint32_t)));
    for (int k = 2; k < 4; ++k) HIP_TRY(hipMalloc((void**)&v.shadowQueues[k], cap * (size_t)(requests ? requests : 1u) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&v.counts, (size_t)RT_VCM_NUM_COUNT_PLANES * RT_VCM_COUNT_PLANE * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&v.passDev, (size_t)RT_VCM_MAX_BATCH * sizeof(DevPass)));
    HIP_TRY(hipMalloc((void**)&v.seedDev, (size_t)RT_VCM_MAX_BATCH * RTGPU_MAX_DIMENSIONS * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&v.devsDev, (size_t)RT_VCM_MAX_BATCH * sizeof(VcmDev)));
    HIP_TRY(hipMalloc((void**)&v.gridsDev, (size_t)RT_VCM_MAX_BATCH * sizeof(HashGridView)));
    v.requestsPerVertex = requests;
    v.batchCapacity = batch;
    v.havePhotons = false;
    return RTGPU_OK;
}

static void launchTrace(RtgpuContext* c, hipStream_t stream, const Paths& paths, const uint32_t* tq, const uint32_t* tqc, const uint32_t* tsq, const uint32_t* tsc, uint32_t* cursor,
                        float shadowOffset = 0.0001f, uint32_t* overflowQueue = nullptr, uint32_t* overflowCount = nullptr)
{
    TravTuning tune = c->tune; tune.shadowOffset = shadowOffset;
    const bool monsters = overflowQueue && tq && !c->countIntersections && c->sceneDev.numObjects == 1u;
    tune.overflowQueue = monsters ? overflowQueue : nullptr; tune.overflowCount = monsters ? overflowCount : nullptr;
    static const int abortEnv = getenv("RTGPU_ABORT_CLOSEST_AFTER") ? atoi(getenv("RTGPU_ABORT_CLOSEST_AFTER")) : -1;   // test hook: 0 sends every ray in flight at exhaustion
    if (abortEnv >= 0) tune.abortClosestAfter = (uint32_t)abortEnv;
    const uint32_t stackClass = c->traversalStackNeed <= 24 ? 24u : (c->traversalStackNeed <= 32 ? 32u : 64u);
    dim3 travGrid(c->numCUs * (c->travBlocksPerCU ? c->travBlocksPerCU : (stackClass == 24u ? 5u : (stackClass == 32u ? 4u : 2u)))), block(RT_BLOCK);
    RtgpuContext::Vcm& v = c->vcm;
    // RTGPU_VCM_WIDE=1: the 4-wide walks in front of the bidirectional integrator's launches too.  Bit-exact (tests/test_gpu_vcm.py), and measured
    // 2 % SLOWER on the Sponza-class scene (16.2 -> 16.6 ms per pass): this pipeline runs on ONE stream, so nothing hides the forty extra re-trace
    // launches per pass batch, and BASELINE config 5 is three analytic objects, which the wide walk does not serve anyway.  Off by default.
    static const bool vcmWide = getenv("RTGPU_VCM_WIDE") && atoi(getenv("RTGPU_VCM_WIDE")) != 0;
    if (vcmWide && useWide(c) && v.exactQueue && v.traceSerial < 2u * RT_VCM_COUNT_PLANE)
    {
        // what the wide walk does not decide goes through the binary-tree kernel below, which keeps its hand-over of degenerate closest-hit rays to
        // k_trace_monster
        const uint32_t k = v.traceSerial++;
        uint32_t* exactCount = v.counts + 10u * RT_VCM_COUNT_PLANE + k; uint32_t* exactShadowCount = v.counts + 12u * RT_VCM_COUNT_PLANE + k;
        launchTraceWide(c, stream, paths, tq, tqc, tsq, tsc, cursor, v.exactQueue, exactCount, v.exactShadowQueue, exactShadowCount, shadowOffset, nullptr, 0u, false);
        tq = v.exactQueue; tqc = exactCount; tsq = v.exactShadowQueue; tsc = exactShadowCount; cursor = v.counts + 14u * RT_VCM_COUNT_PLANE + k;
        travGrid = dim3(c->numCUs);
    }
    LaunchTimer t(c, stream, KC_TRACE);
#define RT_VCM_TRACE(S, C) hipLaunchKernelGGL((k_trace<S, C>), travGrid, block, 0, stream, c->sceneDev, paths, tq, tqc, tsq, tsc, cursor, c->counters, tune)
    if (stackClass == 24u) { if (c->countIntersections) RT_VCM_TRACE(24, true); else RT_VCM_TRACE(24, false); }
    else if (stackClass == 32u) { if (c->countIntersections) RT_VCM_TRACE(32, true); else RT_VCM_TRACE(32, false); }
    else { if (c->countIntersections) RT_VCM_TRACE(64, true); else RT_VCM_TRACE(64, false); }
#undef RT_VCM_TRACE
    if (monsters) hipLaunchKernelGGL(k_trace_monster, dim3(64), dim3(RT_MONSTER_BLOCK), 0, stream, c->sceneDev, paths, overflowQueue, overflowCount);
}

static DevPass makeDevPass(RtgpuContext* c, const RtPassParams* p, const uint32_t* seedDev, uint32_t maxRayDepth)
{
    DevPass pass; memset(&pass, 0, sizeof(pass));
    pass.camera = p->camera; pass.seed = seedDev; pass.numDimensions = p->numDimensions;
    pass.blueNoiseLayers = (c->sceneDev.blueNoise && p->useBlueNoise) ? 4u : 0u;
    pass.sampleOffset[0] = p->sampleOffset[0]; pass.sampleOffset[1] = p->sampleOffset[1];
    pass.passIndex = p->passIndex; pass.maxRayDepth = maxRayDepth; pass.rngKey[0] = p->rngKey[0]; pass.rngKey[1] = p->rngKey[1];
    pass.width = c->width; pass.height = c->height;
    return pass;
}

// VertexConnectionAndMerging::PreRender(passNumber, film), .cpp:84-124, for the next pass in sequence
static VcmDev vcmPreRender(RtgpuContext* c, uint32_t passNumber)
{
    RtgpuContext::Vcm& v = c->vcm;
    const RtVcmParams& vp = v.params;
    const uint32_t lightPathsCount = c->height * c->width;
    if (passNumber == 0u) { v.mergingRadiusVC = vp.initialMergingRadius; v.mergingRadiusVM = vp.initialMergingRadius; }
    else
    {
        v.mergingRadiusVM = v.mergingRadiusVC;
        v.mergingRadiusVC *= vp.mergingRadiusMultiplier;
        v.mergingRadiusVC = v.mergingRadiusVC > vp.minMergingRadius ? v.mergingRadiusVC : vp.minMergingRadius;
    }
    VcmDev dev; memset(&dev, 0, sizeof(dev));
    dev.maxPathLength = vp.maxPathLength; dev.useVertexConnection = vp.useVertexConnection; dev.useVertexMerging = vp.useVertexMerging; dev.iteration = passNumber;
    dev.vertexMergingNormalizationFactor = 1.0f / ((v.mergingRadiusVM * v.mergingRadiusVM) * RTD_PI * lightPathsCount);
    {
        const float etaVCM = RTD_PI * (v.mergingRadiusVC * v.mergingRadiusVC) * lightPathsCount;
        dev.misVertexMergingWeightFactorVC = (vp.useVertexMerging && passNumber > 0u) ? etaVCM : 0.0f;
        dev.misVertexConnectionWeightFactorVC = vp.useVertexConnection ? (1.f / etaVCM) : 0.0f;
    }
    {
        const float etaVCM = RTD_PI * (v.mergingRadiusVM * v.mergingRadiusVM) * lightPathsCount;
        dev.misVertexMergingWeightFactorVM = vp.useVertexMerging ? etaVCM : 0.0f;
        dev.misVertexConnectionWeightFactorVM = vp.useVertexConnection ? (1.f / etaVCM) : 0.0f;
    }
    memcpy(dev.bsdfSamplingWeight, vp.bsdfSamplingWeight, 16); memcpy(dev.lightSamplingWeight, vp.lightSamplingWeight, 16);
    memcpy(dev.vertexConnectingWeight, vp.vertexConnectingWeight, 16); memcpy(dev.cameraConnectingWeight, vp.cameraConnectingWeight, 16);
    memcpy(dev.vertexMergingWeight, vp.vertexMergingWeight, 16);
    return dev;
}

// HashGrid::Build (PreRenderGlobal, .cpp:140-170) over the photons pass `photonPass` of the arena recorded, as the merge set `grid`
static int vcmBuildGrid(RtgpuContext* c, hipStream_t stream, uint32_t photonPass, float radius, VcmPhotonGrid& grid, HashGridView& view)
{
    RtgpuContext::Vcm& v = c->vcm;
    memset(&view, 0, sizeof(view));
    VcmPhotonInput in = { v.arena.photonRaw + (size_t)photonPass * c->numSlots, v.arena.photonCount + (size_t)photonPass * c->numSlots, c->slotPixel,
                          c->numSlots, v.arena.capacity, c->width, c->height, v.arena.maxLV };
    const int e = vcmBuildPhotonGrid(in, radius, stream, grid);
    if (e != 0) return fail(e == (int)hipErrorOutOfMemory ? RTGPU_ERR_OUT_OF_MEMORY : RTGPU_ERR_DEVICE, std::string("photon grid: ") + hipGetErrorString((hipError_t)e));
    view.photons = reinterpret_cast<const Photon*>(grid.sorted); view.cellEnds = grid.cellEnds;
    view.radiusSqr = grid.radiusSqr; view.invCellSize = grid.invCellSize; view.hashTableMask = grid.hashTableMask; view.numPhotons = grid.numPhotons;
    if (grid.numPhotons) HIP_TRY(hipMemcpyAsync(view.boxMin, grid.boxMin, 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return RTGPU_OK;
}

// Submits the queued VertexConnectionAndMerging passes as one batch (launch sequence: rt_vcm.inl)
static int vcmFlush(RtgpuContext* c)
{
    RtgpuContext::Vcm& v = c->vcm;
    if (v.pending.empty()) return RTGPU_OK;
    const RtVcmParams& vp = v.params;
    const uint32_t numPasses = (uint32_t)v.pending.size();
    const uint32_t maxLV = vp.maxPathLength > 1u ? vp.maxPathLength - 1u : 1u;
    int r = RTGPU_OK;
    do
    {
        if (c->shard.rank != 0 || c->shard.worldSize != 1) { r = fail(RTGPU_ERR_UNSUPPORTED, "VCM needs the whole frame on one device (shard {0, 1})"); break; }
        if (!c->activeMask.empty()) { r = fail(RTGPU_ERR_UNSUPPORTED, "VCM does not support active-block restriction"); break; }
        if (c->numLights + maxLV > 64u) { r = fail(RTGPU_ERR_UNSUPPORTED, "VCM: lights + light vertices per pixel must not exceed 64"); break; }
        if ((r = flushPending(c)) != RTGPU_OK) break;
        { const hipError_t e = syncLanes(c); if (e != hipSuccess) { r = fail(RTGPU_ERR_DEVICE, hipGetErrorString(e)); break; } }
        if ((r = ensureVcm(c, maxLV, v.batch > numPasses ? v.batch : numPasses)) != RTGPU_OK) break;
    } while (false);
    if (r) { v.pending.clear(); return r; }
    hipStream_t stream = c->lanes[0].stream;

    // per-pass constants and PreRender state, in pass order; pass 0's merge set comes from the last pass of the previous batch
    // (its per-slot photon storage is about to be overwritten, so that grid is built first)
    std::vector<DevPass> passes(numPasses); std::vector<VcmDev> devs(numPasses); std::vector<HashGridView> grids(numPasses);
    std::vector<float> radiusVM(numPasses);
    for (uint32_t j = 0; j < numPasses; ++j)
    {
        const RtPassParams& p = v.pending[j].params;
        uint32_t* seedDev = v.seedDev + (size_t)j * RTGPU_MAX_DIMENSIONS;
        if (p.numDimensions) HIP_TRY(hipMemcpyAsync(seedDev, v.pending[j].seeds.data(), p.numDimensions * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        passes[j] = makeDevPass(c, &p, seedDev, vp.maxPathLength);
        if (p.passIndex == 0u) v.havePhotons = false;   // (only the first pass of a batch can be a restart: rtgpu_render_pass flushes before it)
        devs[j] = vcmPreRender(c, p.passIndex);
        radiusVM[j] = v.mergingRadiusVM;
        memset(&grids[j], 0, sizeof(HashGridView));
    }
    if (vp.useVertexMerging && v.havePhotons)
        { const int e = vcmBuildGrid(c, stream, v.lastPhotonPass, radiusVM[0], v.grids[0], grids[0]); if (e) { v.pending.clear(); return e; } }
    HIP_TRY(hipMemcpyAsync(v.passDev, passes.data(), numPasses * sizeof(DevPass), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(v.devsDev, devs.data(), numPasses * sizeof(VcmDev), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(v.gridsDev, grids.data(), numPasses * sizeof(HashGridView), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));   // the host vectors and the pending seeds are temporaries
    const VcmBatch batch = { v.passDev, v.devsDev, v.gridsDev, c->numSlots };

    HIP_TRY(hipMemsetAsync(v.counts, 0, (size_t)RT_VCM_NUM_COUNT_PLANES * RT_VCM_COUNT_PLANE * sizeof(uint32_t), stream));
    v.traceSerial = 0u;
    uint32_t* lpc = v.counts; uint32_t* lsc = v.counts + RT_VCM_COUNT_PLANE; uint32_t* lcur = v.counts + 2 * RT_VCM_COUNT_PLANE;
    uint32_t* cpc = v.counts + 3 * RT_VCM_COUNT_PLANE; uint32_t* csc = v.counts + 4 * RT_VCM_COUNT_PLANE; uint32_t* ccur = v.counts + 5 * RT_VCM_COUNT_PLANE;
    uint32_t* cmc = v.counts + 6 * RT_VCM_COUNT_PLANE; uint32_t* ccc = v.counts + 9 * RT_VCM_COUNT_PLANE;
    uint32_t** lq = v.queues; uint32_t** cq = v.queues + 2; uint32_t** lsq = v.shadowQueues; uint32_t** csq = v.shadowQueues + 2;

    const uint32_t totalSlots = c->numSlots * numPasses;
    const uint32_t maxBlocks = c->numCUs * 8u;
    const uint32_t blocksNeeded = (totalSlots + RT_BLOCK - 1) / RT_BLOCK;
    const uint32_t pixelBlocks = (c->numSlots + RT_BLOCK - 1) / RT_BLOCK;
    const dim3 grid1(blocksNeeded < maxBlocks ? blocksNeeded : maxBlocks), pixelGrid(pixelBlocks < maxBlocks ? pixelBlocks : maxBlocks), block(RT_BLOCK);

    {
        LaunchTimer t(c, stream, KC_GENERATE);
        hipLaunchKernelGGL(k_generate, grid1, block, 0, stream, c->sceneDev, v.passDev, c->numSlots, v.cameraPaths, c->slotPixel, totalSlots, cq[0], cpc + 0, c->counters);
        RT_LAUNCH_VCM(k_vcm_emit, grid1, block, 0, stream, c->sceneDev, batch, v.lightPaths, v.cameraPaths, v.arena, c->slotPixel, totalSlots, lq[0], lpc + 0);
    }
    // light sub-paths of every pass of the batch
    for (uint32_t b = 0; b < maxLV; ++b)
    {
        const bool haveShadow = b > 0 && vp.useVertexConnection;
        launchTrace(c, stream, v.lightPaths, lq[b & 1u], lpc + b, haveShadow ? lsq[(b - 1u) & 1u] : nullptr, haveShadow ? lsc + (b - 1u) : nullptr, lcur + b, 0.0001f,
                    v.overflowQueue, v.counts + 7 * RT_VCM_COUNT_PLANE + b);
        LaunchTimer t(c, stream, KC_SHADE);
        RT_LAUNCH_VCM(k_vcm_light_shade, grid1, block, 0, stream, c->sceneDev, batch, v.lightPaths, v.arena, lq[b & 1u], lpc + b, lq[(b + 1u) & 1u], lpc + b + 1,
                           lsq[b & 1u], lsc + b, c->sum, c->secondary, c->counters);
    }
    if (vp.useVertexConnection)
    {
        launchTrace(c, stream, v.lightPaths, nullptr, nullptr, lsq[(maxLV - 1u) & 1u], lsc + (maxLV - 1u), lcur + maxLV);
        LaunchTimer t(c, stream, KC_ACCUMULATE);
        hipLaunchKernelGGL(k_vcm_light_finish, grid1, block, 0, stream, batch, v.lightPaths, totalSlots, c->sum, c->secondary, c->counters);
    }
    // merge sets of passes 1.. of the batch: the photons the pass before them has just recorded
    if (vp.useVertexMerging && numPasses > 1u)
    {
        for (uint32_t j = 1; j < numPasses; ++j)
            { const int e = vcmBuildGrid(c, stream, j - 1u, radiusVM[j], v.grids[j], grids[j]); if (e) { v.pending.clear(); return e; } }
        HIP_TRY(hipMemcpyAsync(v.gridsDev, grids.data(), numPasses * sizeof(HashGridView), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    bool anyPhotons = false;
    for (uint32_t j = 0; j < numPasses; ++j) anyPhotons = anyPhotons || (grids[j].numPhotons != 0u && passes[j].passIndex > 0u);
    // camera sub-paths
    static const uint32_t mergeCooperativeMin = getenv("RTGPU_VCM_MERGE_COOP") ? (uint32_t)atoi(getenv("RTGPU_VCM_MERGE_COOP")) : RT_VCM_COOPERATIVE_MERGE_MIN;   // tuning knob
    for (uint32_t d = 0; d < vp.maxPathLength; ++d)
    {
        const bool haveShadow = d > 0;
        launchTrace(c, stream, v.cameraPaths, cq[d & 1u], cpc + d, haveShadow ? csq[(d - 1u) & 1u] : nullptr, haveShadow ? csc + (d - 1u) : nullptr, ccur + d, 0.0001f,
                    v.overflowQueue, v.counts + 8 * RT_VCM_COUNT_PLANE + d);
        LaunchTimer t(c, stream, KC_SHADE);
        RT_LAUNCH_VCM(k_vcm_camera_shade, grid1, block, 0, stream, c->sceneDev, batch, v.cameraPaths, v.arena, cq[d & 1u], cpc + d, cq[(d + 1u) & 1u], cpc + d + 1,
                           csq[d & 1u], csc + d, v.mergeQueue, cmc + d, v.connectQueue, ccc + d, c->counters);
        if (vp.useVertexConnection && maxLV > 0u && d + 1u < vp.maxPathLength)
            hipLaunchKernelGGL(k_vcm_connect, grid1, block, 0, stream, c->sceneDev, batch, v.cameraPaths, v.arena, v.connectQueue, ccc + d, csq[d & 1u], csc + d);
        if (anyPhotons && vp.useVertexMerging)
            hipLaunchKernelGGL(k_vcm_merge, grid1, block, 0, stream, c->sceneDev, batch, v.arena, v.mergeQueue, cmc + d, mergeCooperativeMin);
    }
    launchTrace(c, stream, v.cameraPaths, nullptr, nullptr, csq[(vp.maxPathLength - 1u) & 1u], csc + (vp.maxPathLength - 1u), ccur + vp.maxPathLength);
    {
        LaunchTimer t(c, stream, KC_ACCUMULATE);
        hipLaunchKernelGGL(k_vcm_camera_finish, pixelGrid, block, 0, stream, batch, numPasses, v.cameraPaths, v.arena, c->sum, c->secondary, c->width, c->counters);
    }
    HIP_TRY(hipGetLastError());
    v.havePhotons = vp.useVertexMerging != 0u;
    v.lastPhotonPass = numPasses - 1u;
    v.pending.clear();
    return RTGPU_OK;
}

// One VertexConnectionAndMerging pass: queued; submitted when the batch is full or anything synchronises
static int vcmRenderPass(RtgpuContext* c, const RtPassParams* p)
{
    RtgpuContext::Vcm& v = c->vcm;
    if (c->shard.rank != 0 || c->shard.worldSize != 1) return fail(RTGPU_ERR_UNSUPPORTED, "VCM needs the whole frame on one device (shard {0, 1})");
    if (!c->activeMask.empty()) return fail(RTGPU_ERR_UNSUPPORTED, "VCM does not support active-block restriction");
    if (c->numLights + (v.params.maxPathLength > 1u ? v.params.maxPathLength - 1u : 1u) > 64u) return fail(RTGPU_ERR_UNSUPPORTED, "VCM: lights + light vertices per pixel must not exceed 64");
    if (p->passIndex == 0u && !v.pending.empty()) { const int r = vcmFlush(c); if (r) return r; }   // a restart begins its own batch
    if (v.pending.empty())
    {
        // passes per launch sequence: enough to keep 256 CUs busy through the tails of every bounce, bounded by the arena footprint
        // (about 2 KB per pixel and pass at path length 10)
        static const int batchEnv = getenv("RTGPU_VCM_BATCH") ? atoi(getenv("RTGPU_VCM_BATCH")) : 0;
        uint32_t batch = batchEnv > 0 ? (uint32_t)batchEnv : 8u;
        if (batch > RT_VCM_MAX_BATCH) batch = RT_VCM_MAX_BATCH;
        const uint32_t maxLV = v.params.maxPathLength > 1u ? v.params.maxPathLength - 1u : 1u;
        const size_t perSlot = ((size_t)2 * R_NUM_BASE + RT_SHADOW_RECORDS * (1u + c->numLights + maxLV) + V_NUM + (size_t)maxLV * (RT_VCM_LV_RECORDS + 2u) + RT_VCM_LV_RECORDS) * sizeof(float4)
                               + (size_t)(10u + c->numLights + maxLV) * 2u * sizeof(uint32_t);
        while (batch > 1u && perSlot * c->numSlots * batch > ((size_t)48 << 30)) --batch;
        v.batch = batch;
    }
    RtgpuContext::Vcm::Pending pd;
    pd.params = *p; pd.params.seed = nullptr;
    pd.seeds.assign(p->seed, p->seed + p->numDimensions);
    v.pending.push_back(std::move(pd));
    if (v.pending.size() >= v.batch) return vcmFlush(c);
    return RTGPU_OK;
}

// One LightTracer pass (Core/Rendering/LightTracer.cpp): the VCM light stage without MIS, light vertices and photons
static int lightTracerRenderPass(RtgpuContext* c, const RtPassParams* p)
{
    RtgpuContext::Vcm& v = c->vcm;
    if (c->shard.rank != 0 || c->shard.worldSize != 1) return fail(RTGPU_ERR_UNSUPPORTED, "the Light Tracer needs the whole frame on one device (shard {0, 1})");
    if (!c->activeMask.empty()) return fail(RTGPU_ERR_UNSUPPORTED, "the Light Tracer does not support active-block restriction");
    if (p->maxRayDepth + 2u > RT_VCM_COUNT_PLANE) return fail(RTGPU_ERR_UNSUPPORTED, "Light Tracer: maxRayDepth must be <= 18");
    { int r = flushPending(c); if (r) return r; }
    HIP_TRY(syncLanes(c));
    { int r = ensureVcm(c, 1u, 1u); if (r) return r; }
    hipStream_t stream = c->lanes[0].stream;
    VcmDev dev; memset(&dev, 0, sizeof(dev));
    dev.maxPathLength = p->maxRayDepth;   // k_lt_shade reads it as RenderingParams::maxRayDepth
    const DevPass pass = makeDevPass(c, p, v.seedDev, p->maxRayDepth);
    HashGridView noGrid; memset(&noGrid, 0, sizeof(noGrid));
    if (p->numDimensions) HIP_TRY(hipMemcpyAsync(v.seedDev, p->seed, p->numDimensions * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(v.passDev, &pass, sizeof(pass), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(v.devsDev, &dev, sizeof(dev), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(v.gridsDev, &noGrid, sizeof(noGrid), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const VcmBatch batch = { v.passDev, v.devsDev, v.gridsDev, c->numSlots };
    HIP_TRY(hipMemsetAsync(v.counts, 0, (size_t)RT_VCM_NUM_COUNT_PLANES * RT_VCM_COUNT_PLANE * sizeof(uint32_t), stream));
    v.traceSerial = 0u;
    uint32_t* lpc = v.counts; uint32_t* lsc = v.counts + RT_VCM_COUNT_PLANE; uint32_t* lcur = v.counts + 2 * RT_VCM_COUNT_PLANE;
    uint32_t* cpc = v.counts + 3 * RT_VCM_COUNT_PLANE;
    uint32_t** lq = v.queues; uint32_t** lsq = v.shadowQueues;
    const uint32_t maxBlocks = c->numCUs * 8u;
    const uint32_t blocksNeeded = (c->numSlots + RT_BLOCK - 1) / RT_BLOCK;
    const dim3 grid1(blocksNeeded < maxBlocks ? blocksNeeded : maxBlocks), block(RT_BLOCK);
    {
        LaunchTimer t(c, stream, KC_GENERATE);
        // the camera ray is generated (it consumes the pixel's lens samples and counts as a primary ray) and then ignored, Viewport.cpp:305-331
        hipLaunchKernelGGL(k_generate, grid1, block, 0, stream, c->sceneDev, v.passDev, c->numSlots, v.cameraPaths, c->slotPixel, c->numSlots, v.queues[2], cpc + 0, c->counters);
        RT_LAUNCH_VCM(k_vcm_emit, grid1, block, 0, stream, c->sceneDev, batch, v.lightPaths, v.cameraPaths, v.arena, c->slotPixel, c->numSlots, lq[0], lpc + 0);
    }
    for (uint32_t b = 0; b <= p->maxRayDepth; ++b)
    {
        const bool haveShadow = b > 0;
        launchTrace(c, stream, v.lightPaths, lq[b & 1u], lpc + b, haveShadow ? lsq[(b - 1u) & 1u] : nullptr, haveShadow ? lsc + (b - 1u) : nullptr, lcur + b, 0.0f,
                    v.overflowQueue, v.counts + 7 * RT_VCM_COUNT_PLANE + b);
        LaunchTimer t(c, stream, KC_SHADE);
        RT_LAUNCH_VCM(k_lt_shade, grid1, block, 0, stream, c->sceneDev, batch, v.lightPaths, v.arena, lq[b & 1u], lpc + b, lq[(b + 1u) & 1u], lpc + b + 1,
                           lsq[b & 1u], lsc + b, c->sum, c->secondary, c->counters);
    }
    launchTrace(c, stream, v.lightPaths, nullptr, nullptr, lsq[p->maxRayDepth & 1u], lsc + p->maxRayDepth, lcur + p->maxRayDepth + 1u, 0.0f);
    {
        LaunchTimer t(c, stream, KC_ACCUMULATE);
        hipLaunchKernelGGL(k_vcm_light_finish, grid1, block, 0, stream, batch, v.lightPaths, c->numSlots, c->sum, c->secondary, c->counters);
    }
    HIP_TRY(hipGetLastError());
    v.havePhotons = false;
    return RTGPU_OK;
}

