// Camera (reference: Core/Scene/Camera.h).  Ray generation runs on the device; the host keeps the
// parameters and hands them over as RtCamera.
#pragma once

#include "../Math/Math.h"
#include "../../../../include/rtgpu.h"

namespace rt {

enum class BokehShape : uint8 { Circle = 0, Hexagon, Square, NGon, Texture };

struct DOFSettings
{
    float focalPlaneDistance = 2.0f;
    float aperture = 0.1f;
    bool enable = false;
    BokehShape bokehShape = BokehShape::Circle;
    uint32 apertureBlades = 5;
};

class RAYLIB_API Camera
{
public:
    Camera();
    void SetTransform(const math::Transform& transform);
    void SetPerspective(float aspectRatio, float FoV);
    const math::Transform& GetTransform() const { return mTransform; }
    const math::Matrix4& GetLocalToWorld() const { return mLocalToWorld; }

    // false if the configuration needs a feature outside the device scope (non-circular bokeh, barrel distortion)
    bool GetDesc(RtCamera& out) const;

    math::Transform mTransform;
    float mAspectRatio;
    float mFieldOfView;
    DOFSettings mDOF;
    float barrelDistortionConstFactor;
    float barrelDistortionVariableFactor;
    bool enableBarellDistortion;

    // mLocalToWorld.FastInverseNoScale() * Matrix4::MakePerspective(aspect, FoV, 0.01f, 1000.0f) in the reference's
    // operation order (Matrix4.h:186-193, Matrix4.cpp:15-24 and :35-70)
    RAYLIB_API static math::Matrix4 ComputeWorldToScreen(const math::Matrix4& localToWorld, float aspectRatio, float tanHalfFoV);

private:
    float mTanHalfFoV;
    math::Matrix4 mLocalToWorld;
    math::Matrix4 mWorldToScreen;   // as SetPerspective left it (reference Camera.cpp:39-48): not refreshed by SetTransform
};

} // namespace rt
