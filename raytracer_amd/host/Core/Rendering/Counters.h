// Ray counters (reference: Core/Rendering/Counters.h:36-93), intersection counters always present.
#pragma once

#include "../Math/Math.h"

namespace rt {

struct RayTracingCounters
{
    uint64 numRays = 0;            // sum over paths of (depth + 1): "paths x bounces"
    uint64 numShadowRays = 0;
    uint64 numShadowRaysHit = 0;
    uint64 numPrimaryRays = 0;
    uint64 numRayBoxTests = 0;
    uint64 numPassedRayBoxTests = 0;
    uint64 numRayTriangleTests = 0;
    uint64 numPassedRayTriangleTests = 0;
    uint64 numMeshHits = 0;
    uint64 numAnalyticHits = 0;
    uint64 numShadowRayBoxTests = 0;       // not counted by the reference (see include/rtgpu.h)
    uint64 numShadowRayTriangleTests = 0;

    void Reset() { *this = RayTracingCounters(); }
    void Append(const RayTracingCounters& o)
    {
        numRays += o.numRays; numShadowRays += o.numShadowRays; numShadowRaysHit += o.numShadowRaysHit;
        numPrimaryRays += o.numPrimaryRays; numRayBoxTests += o.numRayBoxTests; numPassedRayBoxTests += o.numPassedRayBoxTests;
        numRayTriangleTests += o.numRayTriangleTests; numPassedRayTriangleTests += o.numPassedRayTriangleTests;
        numMeshHits += o.numMeshHits; numAnalyticHits += o.numAnalyticHits;
        numShadowRayBoxTests += o.numShadowRayBoxTests; numShadowRayTriangleTests += o.numShadowRayTriangleTests;
    }
};

} // namespace rt
