// rt_vcm_state.h -- per-slot state and per-batch constants of the bidirectional integrator's kernels (rt_vcm.inl), shared with the host side
#pragma once
#include "rt_device_state.h"
#include "rt_device_vcm.h"
#include "rt_vcm_photons.h"

// ---- per-slot state beyond the PathTracerMIS records ---------------------------------------------------------------------
enum VcmRecord : uint32_t
{
    V_MIS,      // dVC, dVM, dVCM | path length (bits 0-7), isFiniteLight << 9
    V_SIMD0,    // Random::mSeedSimd4[0] (two 64-bit lanes)
    V_SIMD1,    // Random::mSeedSimd4[1]
    V_MERGE,    // camera stage: throughput * vertexMergingColor (rgb) of the vertex whose terms are pending | 1 = present
    V_NUM
};
#define RT_VCM_LV_RECORDS 6u   // light vertex: {pos | material, pathLength << 24}, {tangent | roughness}, {normal | metalness},
                               //               {outgoing dir | dVC}, {baseColor}, {throughput | dVCM}
struct VcmArena
{
    float4* recs;            // V_NUM x capacity
    float4* lightVertices;   // maxLV x RT_VCM_LV_RECORDS x capacity
    float4* photonRaw;       // maxLV x 2 x capacity (this pass's photons, per slot)
    uint32_t* lvCount;       // light vertices of the slot's light sub-path
    uint32_t* photonCount;
    float4* cameraVertex;    // RT_VCM_LV_RECORDS x capacity: the camera vertex whose merge query is pending (w of record 3 = dVM)
    uint32_t capacity, maxLV;
};
RT_DEV float4& vrec(const VcmArena& a, uint32_t record, uint32_t slot) { return a.recs[(size_t)record * a.capacity + slot]; }
RT_DEV float4& cvrec(const VcmArena& a, uint32_t record, uint32_t slot) { return a.cameraVertex[(size_t)record * a.capacity + slot]; }
RT_DEV float4& lvrec(const VcmArena& a, uint32_t vertex, uint32_t record, uint32_t slot) { return a.lightVertices[((size_t)vertex * RT_VCM_LV_RECORDS + record) * a.capacity + slot]; }

struct VcmDev   // VertexConnectionAndMerging members after PreRender (.cpp:84-124)
{
    uint32_t maxPathLength, useVertexConnection, useVertexMerging, iteration;
    float misVertexMergingWeightFactorVC, misVertexConnectionWeightFactorVC, misVertexMergingWeightFactorVM, misVertexConnectionWeightFactorVM;
    float vertexMergingNormalizationFactor;
    float bsdfSamplingWeight[4], lightSamplingWeight[4], vertexConnectingWeight[4], cameraConnectingWeight[4], vertexMergingWeight[4];
};

// A batch of passes rides through one launch sequence (slot = passInBatch * slotsPerPass + pixelSlot): every pass has its own constants,
// its own MIS factors (they depend on the pass number) and its own merge set (the photons of the pass before it).
struct VcmBatch { const DevPass* passes; const VcmDev* vcms; const HashGridView* grids; uint32_t slotsPerPass; };
#define RT_VCM_COOPERATIVE_MERGE_MIN 128u   // measured plateau 128-256 (profiles/r01_tuning_sweep.txt)
