// rt_multi.inl -- one context over several devices of a node (rtgpu_create_multi).  Included by rt_runtime.hip.
//
// The reference scales a frame by handing 2-D tiles to the threads of its pool (Viewport.cpp:244-262, ThreadPool.cpp:176-260); here the
// 64x64 tiles are dealt round-robin to the devices (RtgpuShard: tile % worldSize == rank), each device runs the whole launch sequence over
// its own tiles against its own copy of the scene, and nothing is exchanged while passes render.  The only exchange is at read-back:
// device 0 pulls the peers' tiles into its sum buffers -- one kernel that reads the peers' HBM over xGMI (peer access), or, where the
// devices cannot address each other, hipMemcpyPeerAsync into staging buffers and the same kernel over those.  A pixel has exactly one
// owner and the non-owned pixels of every buffer stay zero, so the gathered frame is the single-device frame bit for bit.
//
// The caller drives all devices from one host thread: every call on the multi context repeats itself on the peers first (RT_FAN_OUT), the
// passes are queued asynchronously on each device's batch lanes, and rtgpu_synchronize submits every device's leftover batch before it
// waits for the first.

#define RTGPU_MAX_DEVICES 16

#define RT_FAN_OUT(ctx, call)                                                                                       \
    do {                                                                                                            \
        for (RtgpuContext* peer : (ctx)->peers) { const int fanResult_ = (call); if (fanResult_) return fanResult_; } \
    } while (0)

struct PeerFilms
{
    const float* sum[RTGPU_MAX_DEVICES];
    const float* secondary[RTGPU_MAX_DEVICES];
};

// one thread per pixel: the owner's three sum floats and three secondary floats; rank 0's own tiles are in place already
__global__ void __launch_bounds__(RT_BLOCK) k_gather_tiles(float* __restrict__ sum, float* __restrict__ secondary, const PeerFilms films, uint32_t width, uint32_t height,
                                                           uint32_t worldSize)
{
    const uint32_t tilesX = (width + 63u) / 64u;
    const uint32_t total = width * height, stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    {
        const uint32_t y = i / width, x = i - y * width;
        const uint32_t owner = ((y >> 6) * tilesX + (x >> 6)) % worldSize;
        if (owner == 0u) continue;
        const size_t idx = 3 * (size_t)i;
        const float* __restrict__ a = films.sum[owner];
        const float* __restrict__ b = films.secondary[owner];
        sum[idx + 0] = a[idx + 0]; sum[idx + 1] = a[idx + 1]; sum[idx + 2] = a[idx + 2];
        secondary[idx + 0] = b[idx + 0]; secondary[idx + 1] = b[idx + 1]; secondary[idx + 2] = b[idx + 2];
    }
}

// Pulls the peers' tiles into c->sum / c->secondary.  Every context must be synchronised (rtgpu_synchronize(c) does all of them).
static int gatherPeers(RtgpuContext* c)
{
    if (c->peers.empty()) return RTGPU_OK;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t world = (uint32_t)c->peers.size() + 1u;
    const size_t floats = (size_t)c->width * c->height * 3;
    hipStream_t st = c->lanes[0].stream;
    const auto gatherStart = std::chrono::steady_clock::now();
    PeerFilms films;
    memset(&films, 0, sizeof(films));
    if (c->stagedGather)
    {
        const size_t need = floats * 2 * c->peers.size();
        if (c->gatherStageFloats < need)
        {
            if (c->gatherStage) { HIP_TRY(hipFree(c->gatherStage)); c->gatherStage = nullptr; c->gatherStageFloats = 0; }
            HIP_TRY(hipMalloc((void**)&c->gatherStage, need * sizeof(float)));
            c->gatherStageFloats = need;
        }
        for (size_t k = 0; k < c->peers.size(); ++k)
        {
            RtgpuContext* p = c->peers[k];
            float* a = c->gatherStage + floats * 2 * k;
            HIP_TRY(hipMemcpyPeerAsync(a, c->device, p->sum, p->device, floats * sizeof(float), st));
            HIP_TRY(hipMemcpyPeerAsync(a + floats, c->device, p->secondary, p->device, floats * sizeof(float), st));
            films.sum[k + 1] = a; films.secondary[k + 1] = a + floats;
        }
    }
    else
        for (size_t k = 0; k < c->peers.size(); ++k) { films.sum[k + 1] = c->peers[k]->sum; films.secondary[k + 1] = c->peers[k]->secondary; }
    const uint32_t pixels = c->width * c->height;
    uint32_t blocks = (pixels + RT_BLOCK - 1) / RT_BLOCK;
    if (blocks > c->numCUs * 16u) blocks = c->numCUs * 16u;
    hipLaunchKernelGGL(k_gather_tiles, dim3(blocks), dim3(RT_BLOCK), 0, st, c->sum, c->secondary, films, c->width, c->height, world);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - gatherStart).count();
    c->multiInfo.gathers++; c->multiInfo.lastGatherMs = ms; c->multiInfo.totalGatherMs += ms;
    return RTGPU_OK;
}
