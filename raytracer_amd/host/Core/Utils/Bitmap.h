// Bitmap: pixel storage with the reference's format enumeration (Core/Utils/Bitmap.h:12-160).  Two uses on this path:
// the R32G32B32_Float sum buffers of the Viewport (tight stride, Bitmap.cpp:97-100) and the texel source of
// BitmapTexture, which the device decodes exactly as Bitmap::GetPixelBlock does (all 23 formats).  Of the file loaders only
// uncompressed BMP is present (DDS / EXR are outside the hot-path scope).
#pragma once
#include <stdio.h>

#include "../Math/Math.h"

namespace rt {

class Bitmap;
using BitmapPtr = std::shared_ptr<Bitmap>;

class RAYLIB_API Bitmap
{
public:
    enum class Format : uint8   // values of the reference's enum = RtBitmapFormat of include/rtgpu.h
    {
        Unknown = 0, R8_UNorm, R8G8_UNorm, B8G8R8_UNorm, B8G8R8A8_UNorm, R8G8B8A8_UNorm, B8G8R8A8_UNorm_Palette, B5G6R5_UNorm,
        R16_UNorm, R16G16_UNorm, R16G16B16A16_UNorm, R32_Float, R32G32_Float, R32G32B32_Float, R32G32B32A32_Float,
        R11G11B10_Float, R16_Half, R16G16_Half, R16G16B16_Half, R16G16B16A16_Half, R9G9B9E5_SharedExp, BC1, BC4, BC5,
    };

    struct InitData
    {
        uint32 width = 0;
        uint32 height = 0;
        Format format = Format::Unknown;
        const void* data = nullptr;   // copied; NULL leaves the pixels zero
        uint32 stride = 0;            // 0 = tight
        bool linearSpace = true;      // false: texels are sRGB, converted on every fetch (Bitmap.cpp:512-516)
        uint32 paletteSize = 0;       // B8G8R8A8_UNorm_Palette: number of palette entries (filled through GetPalette())
    };

    explicit Bitmap(const char* debugName = "<unnamed>") : mDebugName(debugName) {}

    static uint32 BitsPerPixel(Format format);   // 0 for Unknown
    bool Init(const InitData& initData);
    bool Init(uint32 width, uint32 height);      // R32G32B32_Float, zeroed (the Viewport's sum buffers)
    // Uncompressed 24-bit and palette-less 8-bit BMP files, rows as stored, sRGB (Bitmap::LoadBMP, Core/Utils/BitmapBMP.cpp:47-134);
    // the reference's other loaders (DDS, EXR, palettes) are outside the hot-path scope
    bool Load(const char* path);                      // BMP, then DDS (reference: Bitmap::Load tries BMP, DDS, EXR; EXR is not read here)
    bool LoadBMP(FILE* file, const char* path);
    bool LoadDDS(FILE* file, const char* path);      // reference: Core/Utils/BitmapDDS.cpp
    void Clear();
    const char* GetDebugName() const { return mDebugName.c_str(); }
    uint32 GetWidth() const { return mWidth; }
    uint32 GetHeight() const { return mHeight; }
    uint32 GetStride() const { return mStride; }
    Format GetFormat() const { return mFormat; }
    bool IsLinearSpace() const { return mLinearSpace; }
    uint8* GetBytes() { return mData.data(); }
    const uint8* GetBytes() const { return mData.data(); }
    float* GetData() { return reinterpret_cast<float*>(mData.data()); }                // float formats
    const float* GetData() const { return reinterpret_cast<const float*>(mData.data()); }
    size_t GetDataSize() const { return (size_t)mStride * mHeight; }
    uint8* GetPalette() { return mPalette.data(); }
    const uint8* GetPalette() const { return mPalette.data(); }
    uint32 GetPaletteSize() const { return (uint32)(mPalette.size() / 4u); }
    // R32G32B32_Float only (the sum buffers): what the rendering tests read
    const math::Vector4 GetPixel(uint32 x, uint32 y, const bool forceLinearSpace = false) const;
    bool Scale(const math::Vector4& factor);
    // Writes an R32G32B32_Float bitmap (the sum buffers) as an OpenEXR file: channels B, G, R as 32-bit floats, scan lines, no
    // compression (the reference goes through tinyexr with PIZ compression, Core/Utils/BitmapEXR.cpp:196-279; any EXR reader opens
    // either).  false for other formats and when the file cannot be created.
    bool SaveEXR(const char* path, const float exposure = 1.0f) const;
    // raw little-endian dump: "RTF3" magic, width, height (uint32 each), then width*height*3 floats
    bool SaveRaw(const char* path) const;
private:
    std::string mDebugName;
    std::vector<uint8> mData, mPalette;
    uint32 mWidth = 0, mHeight = 0, mStride = 0;
    Format mFormat = Format::Unknown;
    bool mLinearSpace = true;
};

} // namespace rt
