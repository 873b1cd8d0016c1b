// rt_vcm_photons.hip -- HashGrid::Build (reference: Core/Utils/HashGrid.h:17-71) on the device, deterministic.
//
// The reference appends photons to per-thread lists while tiles render and concatenates the lists before the next pass; its
// grid is a counting sort of photon indices by hashed cell.  Here the photons of a pass sit in per-pixel storage and the
// merge set is built in a fixed order, so that the float sums of MergeVertices do not depend on the schedule:
//   1. per-pixel photon counts in ROW-MAJOR pixel order -> exclusive scan -> each photon's index (pixel, then path vertex)
//   2. bounding-box minimum (an exact min, order-free)
//   3. cell of every photon (Teschner hash of the truncated cell coordinates, :148-167), STABLE radix sort of the photon
//      indices by cell = the order the reference's sequential fill produces; cell end offsets by histogram + inclusive scan
#include "rt_vcm_photons.h"
#include <hipcub/hipcub.hpp>
#include <float.h>

#define VP_BLOCK 256

__global__ void __launch_bounds__(VP_BLOCK) k_photon_pixel_counts(VcmPhotonInput in, uint32_t* __restrict__ pixelCounts)
{
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= in.numSlots) return;
    const uint32_t pix = in.slotPixel[slot];
    pixelCounts[(pix >> 16) * in.width + (pix & 0xFFFFu)] = in.countPerSlot[slot];
}

__device__ __forceinline__ void atomicMinFloat(float* addr, float v)
{
    // exact minimum for any sign mix: non-negative floats order like signed ints, negative ones inversely like unsigned ints
    if (v >= 0.0f) atomicMin((int*)addr, __float_as_int(v));
    else atomicMax((unsigned int*)addr, __float_as_uint(v));
}

__global__ void __launch_bounds__(VP_BLOCK) k_photon_compact(VcmPhotonInput in, const uint32_t* __restrict__ pixelOffsets, float4* __restrict__ photons,
                                                             float* __restrict__ boxMin)
{
    __shared__ float sMin[3];
    if (threadIdx.x < 3) sMin[threadIdx.x] = FLT_MAX;
    __syncthreads();
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot < in.numSlots)
    {
        const uint32_t pix = in.slotPixel[slot];
        const uint32_t base = pixelOffsets[(pix >> 16) * in.width + (pix & 0xFFFFu)];
        const uint32_t n = in.countPerSlot[slot];
        for (uint32_t k = 0; k < n; ++k)
        {
            const float4 a = in.raw[(size_t)(k * 2 + 0) * in.capacity + slot], b = in.raw[(size_t)(k * 2 + 1) * in.capacity + slot];
            photons[2 * (size_t)(base + k) + 0] = a; photons[2 * (size_t)(base + k) + 1] = b;
            atomicMinFloat(&sMin[0], a.x); atomicMinFloat(&sMin[1], a.y); atomicMinFloat(&sMin[2], a.z);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 && sMin[threadIdx.x] != FLT_MAX) atomicMinFloat(&boxMin[threadIdx.x], sMin[threadIdx.x]);
}

__device__ __forceinline__ int32_t cvtT(float f) { return (f >= 2147483648.0f || f < -2147483648.0f || f != f) ? (int32_t)0x80000000 : (int32_t)f; }

__global__ void __launch_bounds__(VP_BLOCK) k_photon_cells(const float4* __restrict__ photons, uint32_t numPhotons, const float* __restrict__ boxMin, float invCellSize,
                                                           uint32_t mask, uint32_t* __restrict__ keys, uint32_t* __restrict__ values, uint32_t* __restrict__ cellCounts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numPhotons) return;
    const float4 p = photons[2 * (size_t)i];
    const float cx = invCellSize * (p.x - boxMin[0]), cy = invCellSize * (p.y - boxMin[1]), cz = invCellSize * (p.z - boxMin[2]);
    const uint32_t cell = (((uint32_t)cvtT(cx) * 73856093u) ^ ((uint32_t)cvtT(cy) * 19349663u) ^ ((uint32_t)cvtT(cz) * 83492791u)) & mask;
    keys[i] = cell; values[i] = i;
    atomicAdd(&cellCounts[cell], 1u);
}

// photons in hash-grid order: the range query then walks contiguous memory instead of chasing mIndices
__global__ void __launch_bounds__(VP_BLOCK) k_photon_gather(const float4* __restrict__ photons, const uint32_t* __restrict__ indices, uint32_t numPhotons, float4* __restrict__ sorted)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= numPhotons) return;
    const uint32_t i = indices[j];
    sorted[2 * (size_t)j + 0] = photons[2 * (size_t)i + 0]; sorted[2 * (size_t)j + 1] = photons[2 * (size_t)i + 1];
}

static uint32_t nextPowerOfTwo(uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++; return v; }   // Math.h:235-245

void vcmFreePhotonGrid(VcmPhotonGrid& g)
{
    void* ptrs[] = { g.photons, g.sorted, g.cellEnds, g.boxMin, g.pixelCounts, g.pixelOffsets, g.keys[0], g.keys[1], g.values[0], g.values[1], g.temp };   // indices aliases values[]
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (g.totalHost) (void)hipHostFree(g.totalHost);
    g = VcmPhotonGrid();
}

#define VP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return (int)_e; } while (0)

static int ensureTemp(VcmPhotonGrid& g, size_t bytes)
{
    if (g.tempBytes >= bytes) return 0;
    if (g.temp) (void)hipFree(g.temp);
    g.temp = nullptr; g.tempBytes = 0;
    VP_TRY(hipMalloc(&g.temp, bytes));
    g.tempBytes = bytes;
    return 0;
}

int vcmBuildPhotonGrid(const VcmPhotonInput& in, float radius, hipStream_t stream, VcmPhotonGrid& g)
{
    const size_t numPixels = (size_t)in.width * in.height;
    if (g.pixelCapacity < numPixels + 1)
    {
        if (g.pixelCounts) (void)hipFree(g.pixelCounts);
        if (g.pixelOffsets) (void)hipFree(g.pixelOffsets);
        g.pixelCounts = g.pixelOffsets = nullptr;
        VP_TRY(hipMalloc((void**)&g.pixelCounts, (numPixels + 1) * sizeof(uint32_t)));
        VP_TRY(hipMalloc((void**)&g.pixelOffsets, (numPixels + 1) * sizeof(uint32_t)));
        g.pixelCapacity = numPixels + 1;
    }
    if (!g.totalHost) VP_TRY(hipHostMalloc((void**)&g.totalHost, sizeof(uint32_t)));
    if (!g.boxMin) VP_TRY(hipMalloc((void**)&g.boxMin, 3 * sizeof(float)));

    // 1. photon index = exclusive scan of the per-pixel counts in row-major order (one extra element gives the total)
    VP_TRY(hipMemsetAsync(g.pixelCounts, 0, (numPixels + 1) * sizeof(uint32_t), stream));
    const dim3 block(VP_BLOCK), slotGrid((in.numSlots + VP_BLOCK - 1) / VP_BLOCK);
    if (in.numSlots) hipLaunchKernelGGL(k_photon_pixel_counts, slotGrid, block, 0, stream, in, g.pixelCounts);
    size_t need = 0;
    VP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, g.pixelCounts, g.pixelOffsets, (int)(numPixels + 1), stream));
    { int r = ensureTemp(g, need); if (r) return r; }
    VP_TRY(hipcub::DeviceScan::ExclusiveSum(g.temp, need, g.pixelCounts, g.pixelOffsets, (int)(numPixels + 1), stream));
    VP_TRY(hipMemcpyAsync(g.totalHost, g.pixelOffsets + numPixels, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    VP_TRY(hipStreamSynchronize(stream));
    const uint32_t numPhotons = *g.totalHost;

    g.numPhotons = numPhotons;
    g.radiusSqr = radius * radius;                 // mRadiusSqr = Sqr(radius), HashGrid.h:21
    const float cellSize = radius * 2.0f;          // :22
    g.invCellSize = 1.0f / cellSize;               // :23
    const uint32_t hashTableSize = nextPowerOfTwo(numPhotons);   // :37 (0 photons -> 0 -> mask 0xFFFFFFFF; Process returns early on empty mIndices)
    g.hashTableMask = hashTableSize - 1u;
    if (numPhotons == 0) return 0;

    if (g.photonCapacity < numPhotons)
    {
        void* ptrs[] = { g.photons, g.sorted, g.keys[0], g.keys[1], g.values[0], g.values[1] };
        for (void* p : ptrs) if (p) (void)hipFree(p);
        g.photons = nullptr; g.sorted = nullptr; g.indices = nullptr; g.keys[0] = g.keys[1] = g.values[0] = g.values[1] = nullptr;
        const size_t cap = (size_t)numPhotons + numPhotons / 4 + 1024;
        VP_TRY(hipMalloc((void**)&g.photons, cap * 2 * sizeof(float4)));
        VP_TRY(hipMalloc((void**)&g.sorted, cap * 2 * sizeof(float4)));
        for (int k = 0; k < 2; ++k) { VP_TRY(hipMalloc((void**)&g.keys[k], cap * sizeof(uint32_t))); VP_TRY(hipMalloc((void**)&g.values[k], cap * sizeof(uint32_t))); }
        g.photonCapacity = cap;
    }
    if (g.cellCapacity < hashTableSize)
    {
        if (g.cellEnds) (void)hipFree(g.cellEnds);
        g.cellEnds = nullptr;
        VP_TRY(hipMalloc((void**)&g.cellEnds, (size_t)hashTableSize * 2 * sizeof(uint32_t)));   // [counts | ends]
        g.cellCapacity = hashTableSize;
    }
    uint32_t* cellCounts = g.cellEnds + g.cellCapacity;

    // 2. compact + bounding-box minimum (Box::Empty().min = FLT_MAX, Box.h:40-43; AddPoint = lane-wise min)
    const float init[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    VP_TRY(hipMemcpyAsync(g.boxMin, init, sizeof(init), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_photon_compact, slotGrid, block, 0, stream, in, g.pixelOffsets, g.photons, g.boxMin);

    // 3. cells, stable sort of photon indices by cell, cell end offsets
    VP_TRY(hipMemsetAsync(cellCounts, 0, (size_t)hashTableSize * sizeof(uint32_t), stream));
    const dim3 photonGrid((numPhotons + VP_BLOCK - 1) / VP_BLOCK);
    hipLaunchKernelGGL(k_photon_cells, photonGrid, block, 0, stream, g.photons, numPhotons, g.boxMin, g.invCellSize, g.hashTableMask, g.keys[0], g.values[0], cellCounts);
    int endBit = 1; while (endBit < 32 && (hashTableSize >> endBit) != 0u) ++endBit;
    hipcub::DoubleBuffer<uint32_t> dk(g.keys[0], g.keys[1]), dv(g.values[0], g.values[1]);
    need = 0;
    VP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need, dk, dv, (int)numPhotons, 0, endBit, stream));
    { int r = ensureTemp(g, need); if (r) return r; }
    VP_TRY(hipcub::DeviceRadixSort::SortPairs(g.temp, need, dk, dv, (int)numPhotons, 0, endBit, stream));
    g.indices = dv.Current();
    hipLaunchKernelGGL(k_photon_gather, photonGrid, block, 0, stream, g.photons, g.indices, numPhotons, g.sorted);
    need = 0;
    VP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, need, cellCounts, g.cellEnds, (int)hashTableSize, stream));
    { int r = ensureTemp(g, need); if (r) return r; }
    VP_TRY(hipcub::DeviceScan::InclusiveSum(g.temp, need, cellCounts, g.cellEnds, (int)hashTableSize, stream));
    VP_TRY(hipGetLastError());
    return 0;
}
