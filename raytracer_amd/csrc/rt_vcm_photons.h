// rt_vcm_photons.h -- interface between the integrator kernels (rt_runtime.hip / rt_vcm.inl) and the photon hash-grid
// builder (rt_vcm_photons.hip, the one translation unit that uses hipCUB's device scan / radix sort).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// per-pass photon storage written by the light stage: up to `maxPerPixel` photons per path slot, two float4 per photon,
// photon-major (record r of photon k of slot s at raw[(k * 2 + r) * capacity + s])
struct VcmPhotonInput
{
    const float4* raw; const uint32_t* countPerSlot; const uint32_t* slotPixel;   // slot -> x | y << 16
    uint32_t numSlots, capacity, width, height, maxPerPixel;
};

// the merge set of a pass: photons in (pixel row-major, path vertex) order + the hash grid of Core/Utils/HashGrid.h
struct VcmPhotonGrid
{
    float4* photons = nullptr;       // 2 float4 per photon (32 bytes, VertexConnectionAndMerging::Photon)
    float4* sorted = nullptr;        // the same photons in mIndices order (what the range query reads)
    uint32_t* indices = nullptr;     // HashGrid::mIndices
    uint32_t* cellEnds = nullptr;    // HashGrid::mCellEnds (after Build: the END offset of every cell)
    float* boxMin = nullptr;         // 3 floats (device)
    uint32_t numPhotons = 0, hashTableMask = 0;
    float radiusSqr = 0.0f, invCellSize = 0.0f;
    // scratch
    uint32_t* pixelCounts = nullptr; uint32_t* pixelOffsets = nullptr; uint32_t* keys[2] = { nullptr, nullptr }; uint32_t* values[2] = { nullptr, nullptr };
    void* temp = nullptr; size_t tempBytes = 0; uint32_t* totalHost = nullptr;
    size_t photonCapacity = 0, pixelCapacity = 0, cellCapacity = 0;
};

// HashGrid::Build over the photons of `in` (radius = mMergingRadiusVM).  Synchronises the stream once (the photon count sizes
// the table).  Returns a hipError_t as int.
int vcmBuildPhotonGrid(const VcmPhotonInput& in, float radius, hipStream_t stream, VcmPhotonGrid& grid);
void vcmFreePhotonGrid(VcmPhotonGrid& grid);
