// BitmapTexture (reference: Core/Textures/BitmapTexture.h:12-43; Evaluate: BitmapTexture.cpp:32-93, on the device)
#pragma once

#include "Texture.h"
#include "../Utils/Bitmap.h"

namespace rt {

enum class BitmapTextureFilter : uint8 { NearestNeighbor = 0, Bilinear = 1, Bilinear_SmoothStep = 2 };

class RAYLIB_API BitmapTexture : public ITexture
{
public:
    BitmapTexture() = default;
    explicit BitmapTexture(const BitmapPtr& bitmap) : mBitmap(bitmap) {}
    const char* GetName() const override { return mBitmap ? mBitmap->GetDebugName() : "<none>"; }
    bool Describe(RtTexture& out, std::vector<uint8>& texels) const override;
    void SetFilter(BitmapTextureFilter filter) { mFilter = filter; }
    BitmapTextureFilter GetFilter() const { return mFilter; }
    const BitmapPtr& GetBitmap() const { return mBitmap; }
private:
    BitmapPtr mBitmap;
    BitmapTextureFilter mFilter = BitmapTextureFilter::Bilinear_SmoothStep;   // BitmapTexture.cpp:18
};

} // namespace rt
