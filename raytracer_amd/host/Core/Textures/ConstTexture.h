// ConstTexture (reference: Core/Textures/ConstTexture.h; Evaluate returns the colour)
#pragma once

#include "Texture.h"

namespace rt {

class RAYLIB_API ConstTexture : public ITexture
{
public:
    explicit ConstTexture(const math::Vector4& color) : mColor(color) {}
    const char* GetName() const override { return "const"; }
    bool Describe(RtTexture& out, std::vector<uint8>& texels) const override;
private:
    math::Vector4 mColor;
};

} // namespace rt
