// CheckerboardTexture (reference: Core/Textures/CheckerboardTexture.h; Evaluate: CheckerboardTexture.cpp:31-40, on the device)
#pragma once

#include "Texture.h"

namespace rt {

class RAYLIB_API CheckerboardTexture : public ITexture
{
public:
    CheckerboardTexture(const math::Vector4& colorA, const math::Vector4& colorB) : mColorA(colorA), mColorB(colorB) {}
    const char* GetName() const override { return "checkerboard"; }
    bool Describe(RtTexture& out, std::vector<uint8>& texels) const override;
private:
    math::Vector4 mColorA, mColorB;
};

} // namespace rt
