// NoiseTexture (reference: Core/Textures/NoiseTexture.h; Evaluate: NoiseTexture.cpp:58-164, simplex noise octaves, on the device)
#pragma once

#include "Texture.h"

namespace rt {

class RAYLIB_API NoiseTexture : public ITexture
{
public:
    NoiseTexture(const math::Vector4& colorA, const math::Vector4& colorB, const uint32 numOctaves = 1) : mColorA(colorA), mColorB(colorB), mNumOctaves(numOctaves) {}
    const char* GetName() const override { return "noise"; }
    bool Describe(RtTexture& out, std::vector<uint8>& texels) const override;
private:
    math::Vector4 mColorA, mColorB;
    uint32 mNumOctaves;
};

} // namespace rt
