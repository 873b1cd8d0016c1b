// ITexture: what a MaterialParameter, a normal map or the background light can point at (reference:
// Core/Textures/Texture.h:12-36).  Evaluate() runs on the device; the host object only describes itself for upload.
#pragma once

#include "../Math/Math.h"

struct RtTexture;

namespace rt {

class ITexture
{
public:
    virtual ~ITexture() = default;
    virtual const char* GetName() const = 0;
    // fills the device descriptor; bitmap textures append their rows to `texels` and record the offset
    virtual bool Describe(RtTexture& out, std::vector<uint8>& texels) const = 0;
};
using TexturePtr = std::shared_ptr<ITexture>;

} // namespace rt
