// MixTexture (reference: Core/Textures/MixTexture.h; Evaluate = Lerp(A, B, mask), MixTexture.cpp:23-30, on the device)
#pragma once

#include "Texture.h"

namespace rt {

class RAYLIB_API MixTexture : public ITexture
{
public:
    MixTexture(const TexturePtr& textureA, const TexturePtr& textureB, const TexturePtr& textureMask) : mTextureA(textureA), mTextureB(textureB), mTextureMask(textureMask) {}
    const char* GetName() const override { return "mix"; }
    bool Describe(RtTexture& out, std::vector<uint8>& texels) const override;
    const TexturePtr& GetTextureA() const { return mTextureA; }
    const TexturePtr& GetTextureB() const { return mTextureB; }
    const TexturePtr& GetTextureMask() const { return mTextureMask; }
private:
    TexturePtr mTextureA, mTextureB, mTextureMask;
};

} // namespace rt
