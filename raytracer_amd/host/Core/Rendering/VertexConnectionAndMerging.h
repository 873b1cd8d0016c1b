// Vertex Connection and Merging on the GPU through include/rtgpu.h (RT_INTEGRATOR_VCM)
// (reference: Core/Rendering/VertexConnectionAndMerging.h -- same public "for debugging" members, same defaults).
#pragma once

#include "PathTracerMIS.h"

namespace rt {

class RAYLIB_API VertexConnectionAndMerging : public PathTracerMIS
{
public:
    explicit VertexConnectionAndMerging(const Scene& scene);
    const char* GetName() const override;
    bool RenderPass(const RtPassParams& params) override;

    // for debugging (mBSDFSamplingWeight and mLightSamplingWeight are inherited)
    math::Vector4 mVertexConnectingWeight;
    math::Vector4 mVertexMergingWeight;
    math::Vector4 mCameraConnectingWeight;

    uint32 mMaxPathLength;
    float mInitialMergingRadius;
    float mMinMergingRadius;
    float mMergingRadiusMultiplier;

    bool mUseVertexConnection;
    bool mUseVertexMerging;

private:
    RtVcmParams mApplied;
    bool mHaveApplied = false;
};

} // namespace rt
