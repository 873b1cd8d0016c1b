// PostprocessParams (reference: Core/Rendering/PostProcess.h:10-28, defaults PostProcess.cpp:6-14); the tone mapping
// itself runs on the device (rtgpu_postprocess).
#pragma once

#include "../Math/Math.h"

namespace rt {

enum class Tonemapper : uint8 { Clamped, Reinhard, JimHejland_RichardBurgessDawson, ACES };   // Core/Color/ColorHelpers.h:78-84

struct PostprocessParams
{
    math::Vector4 colorFilter = math::VECTOR_ONE;
    float exposure = 0.0f;             // exposure in log scale
    float contrast = 0.8f;
    float saturation = 0.98f;
    float ditheringStrength = 0.005f;  // applied after tonemapping
    float bloomFactor = 0.0f;          // bloom multiplier
    Tonemapper tonemapper = Tonemapper::ACES;
};

} // namespace rt
