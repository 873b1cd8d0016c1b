// Minimal float3 bitmap: the layout of the reference's R32G32B32_Float sum buffer
// (tight stride, Core/Utils/Bitmap.cpp:97-100) plus the accessors the rendering tests use.
#pragma once

#include "../Math/Math.h"

namespace rt {

class RAYLIB_API Bitmap
{
public:
    bool Init(uint32 width, uint32 height);
    void Clear();
    uint32 GetWidth() const { return mWidth; }
    uint32 GetHeight() const { return mHeight; }
    uint32 GetStride() const { return mWidth * 3u * (uint32)sizeof(float); }
    float* GetData() { return mData.data(); }
    const float* GetData() const { return mData.data(); }
    size_t GetDataSize() const { return mData.size() * sizeof(float); }
    const math::Vector4 GetPixel(uint32 x, uint32 y, const bool forceLinearSpace = false) const;
    bool Scale(const math::Vector4& factor);
    // raw little-endian dump: "RTF3" magic, width, height (uint32 each), then width*height*3 floats
    bool SaveRaw(const char* path) const;
private:
    std::vector<float> mData;
    uint32 mWidth = 0, mHeight = 0;
};

} // namespace rt
