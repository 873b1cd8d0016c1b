// Camera, Bitmap, Viewport and the renderer factory.  Host side.
#include "../Core/Rendering/Viewport.h"
#include "../Core/Rendering/PathTracerMIS.h"
#include "../Core/Rendering/VertexConnectionAndMerging.h"
#include "../Core/Textures/BitmapTexture.h"
#include "../Core/Textures/CheckerboardTexture.h"
#include "../Core/Textures/ConstTexture.h"
#include "../Core/Textures/NoiseTexture.h"
#include "../Core/Textures/MixTexture.h"
#include "../../../include/rtgpu.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace rt {

using namespace math;

static const uint32 MAX_IMAGE_SIZE = 1u << 16;

// ---------------------------------------------------------------------------------------------------
// Camera
// ---------------------------------------------------------------------------------------------------
Camera::Camera()
    : mAspectRatio(1.0f)
    , mFieldOfView(DegToRad(20.0f))
    , barrelDistortionConstFactor(0.01f)
    , barrelDistortionVariableFactor(0.0f)
    , enableBarellDistortion(false)
    , mTanHalfFoV(tanf(DegToRad(20.0f) * 0.5f))
    , mLocalToWorld(Matrix4::Identity())
    , mWorldToScreen(Matrix4::Identity())
{
}

void Camera::SetTransform(const Transform& transform)
{
    mTransform = transform;
    mLocalToWorld = transform.ToMatrix4();
}

void Camera::SetPerspective(float aspectRatio, float FoV)
{
    mAspectRatio = aspectRatio;
    mFieldOfView = FoV;
    mTanHalfFoV = tanf(mFieldOfView * 0.5f);
    mWorldToScreen = ComputeWorldToScreen(mLocalToWorld, mAspectRatio, mTanHalfFoV);
}

Matrix4 Camera::ComputeWorldToScreen(const Matrix4& l, float aspectRatio, float tanHalfFoV)
{
    // FastInverseNoScale: Transpose3 of the three axes (their w lanes end up as c.y, c.w, c.w), then the negated
    // transform of the translation
    const Vector4 a = l[0], b = l[1], c = l[2], t = l[3];
    Matrix4 inv;
    inv[0] = Vector4(a.x, b.x, c.x, c.y);
    inv[1] = Vector4(a.y, b.y, c.y, c.w);
    inv[2] = Vector4(a.z, b.z, c.z, c.w);
    {
        float r[4];
        for (int k = 0; k < 4; ++k)
        {
            float v = t.x * (&inv[0].x)[k];
            v = fmaf(-t.y, (&inv[1].x)[k], -v);
            v = fmaf(-t.z, (&inv[2].x)[k], v);
            r[k] = v;
        }
        inv[3] = Vector4(r[0], r[1], r[2], r[3]);
    }
    const float nearZ = 0.01f, farZ = 1000.0f;
    const float yScale = 1.0f / tanHalfFoV;
    const float xScale = yScale / aspectRatio;
    const Vector4 p[4] = { Vector4(xScale, 0.0f, 0.0f, 0.0f), Vector4(0.0f, yScale, 0.0f, 0.0f),
                           Vector4(0.0f, 0.0f, farZ / (farZ - nearZ), 1.0f), Vector4(0.0f, 0.0f, -nearZ * farZ / (farZ - nearZ), 0.0f) };
    Matrix4 out;
    for (int i = 0; i < 4; ++i)
    {
        const float* ai = &inv[i].x;
        float r[4];
        for (int k = 0; k < 4; ++k)
        {
            float v = ai[0] * (&p[0].x)[k];
            v = fmaf(ai[1], (&p[1].x)[k], v);
            v = fmaf(ai[2], (&p[2].x)[k], v);
            v = fmaf(ai[3], (&p[3].x)[k], v);
            r[k] = v;
        }
        out[i] = Vector4(r[0], r[1], r[2], r[3]);
    }
    return out;
}

bool Camera::GetDesc(RtCamera& out) const
{
    memset(&out, 0, sizeof(out));
    mLocalToWorld.Store(out.localToWorld);
    out.aspectRatio = mAspectRatio;
    out.tanHalfFoV = mTanHalfFoV;
    out.dofEnable = mDOF.enable ? 1u : 0u;
    out.bokehShape = (uint32_t)mDOF.bokehShape;
    out.focalPlaneDistance = mDOF.focalPlaneDistance;
    out.aperture = mDOF.aperture;
    out.barrelDistortionConstFactor = barrelDistortionConstFactor;
    out.barrelDistortionVariableFactor = barrelDistortionVariableFactor;
    mWorldToScreen.Store(out.worldToScreen);
    if (mDOF.enable && (mDOF.bokehShape == BokehShape::NGon || mDOF.bokehShape == BokehShape::Texture))
    {
        fprintf(stderr, "[rt] ERROR: NGon / texture-shaped bokeh is not supported by the device path\n");
        return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Bitmap
// ---------------------------------------------------------------------------------------------------
uint32 Bitmap::BitsPerPixel(Format format)   // Bitmap::BitsPerPixel, Core/Utils/Bitmap.cpp:17-46
{
    switch (format)
    {
    case Format::R8_UNorm: case Format::B8G8R8A8_UNorm_Palette: case Format::BC5: return 8;
    case Format::R8G8_UNorm: case Format::R16_UNorm: case Format::R16_Half: case Format::B5G6R5_UNorm: return 16;
    case Format::B8G8R8_UNorm: return 24;
    case Format::B8G8R8A8_UNorm: case Format::R8G8B8A8_UNorm: case Format::R16G16_UNorm: case Format::R32_Float: case Format::R16G16_Half:
    case Format::R11G11B10_Float: case Format::R9G9B9E5_SharedExp: return 32;
    case Format::R16G16B16_Half: return 48;
    case Format::R16G16B16A16_UNorm: case Format::R32G32_Float: case Format::R16G16B16A16_Half: return 64;
    case Format::R32G32B32_Float: return 96;
    case Format::R32G32B32A32_Float: return 128;
    case Format::BC1: case Format::BC4: return 4;
    default: return 0;
    }
}

bool Bitmap::Init(const InitData& initData)
{
    const uint32 bits = BitsPerPixel(initData.format);
    if (bits == 0 || initData.width == 0 || initData.height == 0)
    {
        fprintf(stderr, "[rt] ERROR: Invalid bitmap format\n");
        return false;
    }
    const uint32 tight = (uint32)((uint64)initData.width * bits / 8u);   // ComputeDataStride, Bitmap.cpp:97-100 (block formats: bytes per texel row)
    mStride = initData.stride > tight ? initData.stride : tight;   // Max(stride, ComputeDataStride), Bitmap.cpp:245
    mWidth = initData.width; mHeight = initData.height; mFormat = initData.format; mLinearSpace = initData.linearSpace;
    mData.assign((size_t)mStride * mHeight, 0);
    if (initData.data) memcpy(mData.data(), initData.data, mData.size());
    mPalette.assign((size_t)initData.paletteSize * 4u, 0);   // B8G8R8A8 entries, filled by the caller (the reference's Init allocates only)
    return true;
}

bool Bitmap::Load(const char* path)   // Bitmap::Load, Core/Utils/Bitmap.cpp:297-331
{
    FILE* file = fopen(path, "rb");
    if (!file) { fprintf(stderr, "[rt] ERROR: Failed to load source image from file '%s'\n", path); return false; }
    bool ok = LoadBMP(file, path);
    if (!ok)
    {
        fseek(file, 0, SEEK_SET);
        ok = LoadDDS(file, path);
        if (!ok) fprintf(stderr, "[rt] ERROR: Failed to load '%s' - unknown format (BMP and DDS are read; EXR is not)\n", path);
    }
    fclose(file);
    return ok;
}

bool Bitmap::LoadBMP(FILE* file, const char* path)
{
#pragma pack(push, 2)
    struct FileHeader { uint16 bfType; uint32 bfSize; uint16 bfReserved1, bfReserved2; uint32 bfOffBits; } fileHeader;
    struct InfoHeader { uint32 biSize; int32 biWidth, biHeight; uint16 biPlanes, biBitCount; uint32 biCompression, biSizeImage; int32 biXPelsPerMeter, biYPelsPerMeter; uint32 biClrUsed, biClrImportant; } infoHeader;
#pragma pack(pop)
    bool ok = fread(&fileHeader, sizeof(fileHeader), 1, file) == 1 && fileHeader.bfType == 0x4D42 && fread(&infoHeader, sizeof(infoHeader), 1, file) == 1;
    if (ok && (infoHeader.biPlanes != 1 || infoHeader.biCompression != 0)) { fprintf(stderr, "[rt] ERROR: Unsupported BMP format: '%s'\n", path); ok = false; }
    InitData init;
    if (ok)
    {
        if (infoHeader.biBitCount == 24) init.format = Format::B8G8R8_UNorm;
        else if (infoHeader.biBitCount == 8 && infoHeader.biClrUsed > 0) init.format = Format::B8G8R8A8_UNorm_Palette;   // BitmapBMP.cpp:79-82
        else if (infoHeader.biBitCount == 8 && infoHeader.biClrUsed == 0) init.format = Format::R8_UNorm;
        else { fprintf(stderr, "[rt] ERROR: Unsupported BMP bit depth (%u): '%s'\n", (uint32)infoHeader.biBitCount, path); ok = false; }
    }
    if (ok && (infoHeader.biWidth <= 0 || infoHeader.biHeight <= 0)) { fprintf(stderr, "[rt] ERROR: Invalid image size: '%s'\n", path); ok = false; }
    if (ok)
    {
        init.linearSpace = false;
        init.width = (uint32)infoHeader.biWidth; init.height = (uint32)infoHeader.biHeight;
        init.stride = ((uint32)infoHeader.biWidth * BitsPerPixel(init.format) / 8u + 3u) & ~3u;   // BMP rows are multiples of 4 bytes
        // the colour table follows the info header (BitmapBMP.cpp:104-119); more than 256 entries cannot be indexed by a byte
        if (init.format == Format::B8G8R8A8_UNorm_Palette) { if (infoHeader.biClrUsed > 256u) { fprintf(stderr, "[rt] ERROR: Unsupported BMP palette size (%u): '%s'\n", infoHeader.biClrUsed, path); return false; } init.paletteSize = infoHeader.biClrUsed; }
        ok = Init(init);
        if (ok && init.paletteSize > 0 && fread(mPalette.data(), (size_t)init.paletteSize * 4u, 1, file) != 1)
        {
            fprintf(stderr, "[rt] ERROR: Failed to read bitmap palette from file '%s'\n", path);
            return false;
        }
        ok = ok && fseek(file, (long)fileHeader.bfOffBits, SEEK_SET) == 0 && fread(mData.data(), mData.size(), 1, file) == 1;
        if (!ok) fprintf(stderr, "[rt] ERROR: Failed to read bitmap data from file '%s'\n", path);
    }
    return ok;
}

// Bitmap::LoadDDS, Core/Utils/BitmapDDS.cpp:232-540: the header decides one of the texel formats the device decodes; the payload is
// taken as is (height x width x bits / 8 bytes -- which is also the size of the 4x4-block formats).
bool Bitmap::LoadDDS(FILE* file, const char* path)
{
    struct PixelFormat { uint32 size, flags, fourCC, rgbBitCount, rBitMask, gBitMask, bBitMask, aBitMask; };
    struct Header { uint32 magic, size, flags, height, width, pitchOrLinearSize, depth, mipMapCount, reserved1[11]; PixelFormat pixelFormat; uint32 caps[4]; uint32 reserved2; } header;
    struct HeaderDX10 { uint32 dxgiFormat, resourceDimension, miscFlag, arraySize, miscFlags2; } dx10;
    static_assert(sizeof(Header) == 128, "DDS header");
    if (fread(&header, sizeof(header), 1, file) != 1) return false;
    if (header.magic != 0x20534444u) return false;
    InitData init;
    init.linearSpace = true;
    init.width = header.width; init.height = header.height;
    if (init.width < 1 || init.height < 1 || init.width >= 65535u || init.height >= 65535u)
    {
        fprintf(stderr, "[rt] ERROR: Unsupported DDS format in file '%s': dimensions are out of bounds (%ux%u)\n", path, init.width, init.height);
        return false;
    }
    auto fourCC = [](char a, char b, char c, char d) { return (uint32)(uint8)a | ((uint32)(uint8)b << 8) | ((uint32)(uint8)c << 16) | ((uint32)(uint8)d << 24); };
    const PixelFormat& pf = header.pixelFormat;
    if (pf.flags & 0x40u)   // DDPF_RGB
    {
        if (pf.rgbBitCount == 32)
        {
            if (pf.rBitMask == 0x00FF0000u && pf.gBitMask == 0x0000FF00u && pf.bBitMask == 0x000000FFu && pf.aBitMask == 0xFF000000u) { init.format = Format::B8G8R8A8_UNorm; init.linearSpace = false; }
            else if (pf.rBitMask == 0x000000FFu && pf.gBitMask == 0x0000FF00u && pf.bBitMask == 0x00FF0000u && pf.aBitMask == 0xFF000000u) { init.format = Format::R8G8B8A8_UNorm; init.linearSpace = false; }
            else if (pf.rBitMask == 0xFFFFFFFFu && pf.gBitMask == 0 && pf.bBitMask == 0 && pf.aBitMask == 0) init.format = Format::R32_Float;
            else if (pf.rBitMask == 0xFFFFu && pf.gBitMask == 0xFFFF0000u && pf.bBitMask == 0 && pf.aBitMask == 0) init.format = Format::R16G16_UNorm;
        }
        else if (pf.rgbBitCount == 16)
        {
            if (pf.rBitMask == 0xF800u && pf.gBitMask == 0x07E0u && pf.bBitMask == 0x001Fu && pf.aBitMask == 0) { init.format = Format::B5G6R5_UNorm; init.linearSpace = false; }
        }
    }
    else if (pf.flags & 0x4u)   // DDPF_FOURCC
    {
        if (pf.fourCC == 111u) init.format = Format::R16_Half;                    // D3DFMT_R16F ...
        else if (pf.fourCC == 112u) init.format = Format::R16G16_Half;
        else if (pf.fourCC == 113u) init.format = Format::R16G16B16A16_Half;
        else if (pf.fourCC == 114u) init.format = Format::R32_Float;
        else if (pf.fourCC == 115u) init.format = Format::R32G32_Float;
        else if (pf.fourCC == 116u) init.format = Format::R32G32B32A32_Float;
        else if (pf.fourCC == 36u) init.format = Format::R16G16B16A16_UNorm;      // D3DFMT_A16B16G16R16
        else if (pf.fourCC == fourCC('D', 'X', 'T', '1')) init.format = Format::BC1;
        else if (pf.fourCC == fourCC('A', 'T', 'I', '1') || pf.fourCC == fourCC('B', 'C', '4', 'U') || pf.fourCC == fourCC('B', 'C', '4', 'S')) init.format = Format::BC4;
        else if (pf.fourCC == fourCC('A', 'T', 'I', '2') || pf.fourCC == fourCC('B', 'C', '5', 'U')) init.format = Format::BC5;
        else if (pf.fourCC == fourCC('D', 'X', '1', '0'))
        {
            if (fread(&dx10, sizeof(dx10), 1, file) != 1) { fprintf(stderr, "[rt] ERROR: Failed to read DX10 header '%s'\n", path); return false; }
            switch (dx10.dxgiFormat)   // DXGI_FORMAT values
            {
            case 10: init.format = Format::R16G16B16A16_Half; break;
            case 34: init.format = Format::R16G16_Half; break;
            case 54: init.format = Format::R16_Half; break;
            case 2: init.format = Format::R32G32B32A32_Float; break;
            case 6: init.format = Format::R32G32B32_Float; break;
            case 16: init.format = Format::R32G32_Float; break;
            case 41: init.format = Format::R32_Float; break;
            case 67: init.format = Format::R9G9B9E5_SharedExp; break;
            case 26: init.format = Format::R11G11B10_Float; break;
            case 87: init.format = Format::B8G8R8A8_UNorm; break;
            case 91: init.format = Format::B8G8R8A8_UNorm; init.linearSpace = false; break;
            case 49: init.format = Format::R8G8_UNorm; break;
            case 61: init.format = Format::R8_UNorm; break;
            case 85: init.format = Format::B5G6R5_UNorm; init.linearSpace = false; break;
            case 11: init.format = Format::R16G16B16A16_UNorm; break;
            case 35: init.format = Format::R16G16_UNorm; break;
            case 56: init.format = Format::R16_UNorm; break;
            case 71: init.format = Format::BC1; break;
            case 72: init.format = Format::BC1; init.linearSpace = false; break;
            case 80: init.format = Format::BC4; break;
            case 83: init.format = Format::BC5; break;
            default: break;
            }
        }
    }
    else if (pf.flags & 0x20000u)   // DDPF_LUMINANCE
    {
        if (pf.rgbBitCount == 8)
        {
            if (pf.rBitMask == 0xFFu && pf.gBitMask == 0 && pf.bBitMask == 0 && pf.aBitMask == 0) { init.format = Format::R8_UNorm; init.linearSpace = false; }
            else if (pf.rBitMask == 0xFFu && pf.gBitMask == 0 && pf.bBitMask == 0 && pf.aBitMask == 0xFF00u) { init.format = Format::R8G8_UNorm; init.linearSpace = false; }
        }
        else if (pf.rgbBitCount == 16)
        {
            if (pf.rBitMask == 0xFFFFu && pf.gBitMask == 0 && pf.bBitMask == 0 && pf.aBitMask == 0) init.format = Format::R16_UNorm;
            else if (pf.rBitMask == 0xFFu && pf.gBitMask == 0 && pf.bBitMask == 0 && pf.aBitMask == 0xFF00u) { init.format = Format::R8G8_UNorm; init.linearSpace = false; }
        }
    }
    if (init.format == Format::Unknown) { fprintf(stderr, "[rt] ERROR: Unsupported DDS format in file '%s'\n", path); return false; }
    if (!Init(init)) return false;
    if (fread(mData.data(), mData.size(), 1, file) != 1) { fprintf(stderr, "[rt] ERROR: Failed to read bitmap data from file '%s'\n", path); return false; }
    return true;
}

bool Bitmap::Init(uint32 width, uint32 height)
{
    InitData init;
    init.width = width; init.height = height; init.format = Format::R32G32B32_Float;
    if (width == 0 || height == 0) { mWidth = width; mHeight = height; mStride = width * 12u; mFormat = init.format; mData.clear(); return true; }
    return Init(init);
}

void Bitmap::Clear() { std::fill(mData.begin(), mData.end(), (uint8)0); }

const Vector4 Bitmap::GetPixel(uint32 x, uint32 y, const bool) const
{
    if (mFormat != Format::R32G32B32_Float) return Vector4::Zero();
    const float* p = reinterpret_cast<const float*>(mData.data() + (size_t)mStride * y) + 3 * (size_t)x;
    return Vector4(p[0], p[1], p[2], 0.0f);
}

bool Bitmap::Scale(const Vector4& factor)
{
    if (mFormat != Format::R32G32B32_Float) return false;
    for (uint32 y = 0; y < mHeight; ++y)
    {
        float* row = reinterpret_cast<float*>(mData.data() + (size_t)mStride * y);
        for (uint32 x = 0; x < mWidth; ++x) { row[3 * x + 0] *= factor.x; row[3 * x + 1] *= factor.y; row[3 * x + 2] *= factor.z; }
    }
    return true;
}

bool Bitmap::SaveEXR(const char* path, const float exposure) const
{
    if (mFormat != Format::R32G32B32_Float)
    {
        fprintf(stderr, "[rt] ERROR: Bitmap::SaveEXR: Unsupported format\n");
        return false;
    }
    FILE* file = fopen(path, "wb");
    if (!file)
    {
        fprintf(stderr, "[rt] ERROR: Failed to save EXR file '%s'\n", path);
        return false;
    }
    std::vector<uint8> head;
    auto bytes = [&](const void* p, size_t n) { const uint8* b = static_cast<const uint8*>(p); head.insert(head.end(), b, b + n); };
    auto text = [&](const char* t) { bytes(t, strlen(t) + 1); };
    auto i32 = [&](int32 v) { bytes(&v, 4); };
    auto f32 = [&](float v) { bytes(&v, 4); };
    auto attribute = [&](const char* name, const char* type, int32 size) { text(name); text(type); i32(size); };
    i32(20000630); i32(2);                                   // magic, version 2 (scan lines, single part)
    attribute("channels", "chlist", 3 * 18 + 1);
    for (const char* channel : { "B", "G", "R" }) { text(channel); i32(2); head.push_back(0); head.push_back(0); head.push_back(0); head.push_back(0); i32(1); i32(1); }   // FLOAT, linear, sampling 1 x 1
    head.push_back(0);
    attribute("compression", "compression", 1); head.push_back(0);
    attribute("dataWindow", "box2i", 16); i32(0); i32(0); i32((int32)mWidth - 1); i32((int32)mHeight - 1);
    attribute("displayWindow", "box2i", 16); i32(0); i32(0); i32((int32)mWidth - 1); i32((int32)mHeight - 1);
    attribute("lineOrder", "lineOrder", 1); head.push_back(0);
    attribute("pixelAspectRatio", "float", 4); f32(1.0f);
    attribute("screenWindowCenter", "v2f", 8); f32(0.0f); f32(0.0f);
    attribute("screenWindowWidth", "float", 4); f32(1.0f);
    head.push_back(0);
    const size_t lineBytes = 8 + (size_t)mWidth * 12;
    bool ok = fwrite(head.data(), 1, head.size(), file) == head.size();
    for (uint32 y = 0; y < mHeight && ok; ++y) { const uint64 offset = head.size() + (uint64)mHeight * 8 + (uint64)y * lineBytes; ok = fwrite(&offset, 8, 1, file) == 1; }
    std::vector<float> line((size_t)mWidth * 3);
    for (uint32 y = 0; y < mHeight && ok; ++y)
    {
        const float* row = reinterpret_cast<const float*>(mData.data() + (size_t)mStride * y);
        for (uint32 x = 0; x < mWidth; ++x) for (uint32 c = 0; c < 3; ++c) line[(size_t)c * mWidth + x] = exposure * row[3 * x + (2 - c)];   // planes B, G, R
        const int32 header[2] = { (int32)y, (int32)(mWidth * 12) };
        ok = fwrite(header, 4, 2, file) == 2 && fwrite(line.data(), 4, line.size(), file) == line.size();
    }
    fclose(file);
    if (!ok) fprintf(stderr, "[rt] ERROR: Failed to save EXR file '%s'\n", path);
    return ok;
}

// ---------------------------------------------------------------------------------------------------
// Textures: device descriptors
// ---------------------------------------------------------------------------------------------------
bool BitmapTexture::Describe(RtTexture& out, std::vector<uint8>& texels) const
{
    memset(&out, 0, sizeof(out));
    if (!mBitmap || Bitmap::BitsPerPixel(mBitmap->GetFormat()) == 0)
    {
        fprintf(stderr, "[rt] ERROR: bitmap texture '%s' has no pixels\n", GetName());
        return false;
    }
    out.kind = RT_TEXTURE_BITMAP;
    out.format = (uint32)mBitmap->GetFormat();
    out.width = mBitmap->GetWidth(); out.height = mBitmap->GetHeight(); out.stride = mBitmap->GetStride();
    out.linearSpace = mBitmap->IsLinearSpace() ? 1u : 0u;
    out.filter = (uint32)mFilter;
    while (texels.size() % 16u) texels.push_back(0);
    out.dataOffset = texels.size();
    texels.insert(texels.end(), mBitmap->GetBytes(), mBitmap->GetBytes() + mBitmap->GetDataSize());
    if (mBitmap->GetFormat() == Bitmap::Format::B8G8R8A8_UNorm_Palette)
    {
        while (texels.size() % 16u) texels.push_back(0);
        out.paletteOffset = texels.size();
        texels.insert(texels.end(), mBitmap->GetPalette(), mBitmap->GetPalette() + mBitmap->GetPaletteSize() * 4u);
        if (mBitmap->GetPaletteSize() < 256u) texels.insert(texels.end(), (256u - mBitmap->GetPaletteSize()) * 4u, 0);   // any index byte stays in range
    }
    return true;
}

bool NoiseTexture::Describe(RtTexture& out, std::vector<uint8>&) const
{
    memset(&out, 0, sizeof(out));
    out.kind = RT_TEXTURE_NOISE;
    out.numOctaves = mNumOctaves;
    memcpy(out.colorA, &mColorA, 16); memcpy(out.colorB, &mColorB, 16);
    return true;
}

bool MixTexture::Describe(RtTexture& out, std::vector<uint8>&) const
{
    memset(&out, 0, sizeof(out));
    out.kind = RT_TEXTURE_MIX;   // the child indices are filled by Scene::Flatten, which interns the children first
    return mTextureA && mTextureB && mTextureMask;
}

bool CheckerboardTexture::Describe(RtTexture& out, std::vector<uint8>&) const
{
    memset(&out, 0, sizeof(out));
    out.kind = RT_TEXTURE_CHECKERBOARD;
    memcpy(out.colorA, &mColorA, 16); memcpy(out.colorB, &mColorB, 16);
    return true;
}

bool ConstTexture::Describe(RtTexture& out, std::vector<uint8>&) const
{
    memset(&out, 0, sizeof(out));
    out.kind = RT_TEXTURE_CONST;
    memcpy(out.colorA, &mColor, 16);
    return true;
}

bool Bitmap::SaveRaw(const char* path) const
{
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    const uint32 header[3] = { 0x33465452u /* "RTF3" */, mWidth, mHeight };
    bool ok = fwrite(header, sizeof(header), 1, f) == 1;
    ok = ok && mFormat == Format::R32G32B32_Float && (mData.empty() || fwrite(mData.data(), mData.size(), 1, f) == 1);
    fclose(f);
    return ok;
}

// ---------------------------------------------------------------------------------------------------
// Viewport
// ---------------------------------------------------------------------------------------------------
Viewport::Viewport()
{
    Entropy entropy;
    mRngKey[0] = ((uint64)entropy.GetInt() << 32) | entropy.GetInt();
    mRngKey[1] = ((uint64)entropy.GetInt() << 32) | entropy.GetInt();
}

Viewport::~Viewport() { PinSumBuffers(false); }

// Viewport::GetSumBuffer is a read-back over PCIe here: into page-locked memory it runs at the link's rate (24.9 MB of a 1080p frame:
// ~0.6 ms instead of ~3 ms into pageable memory).  The renderer registers the bitmaps on its device at the first read-back; they are
// released before the bitmaps are re-allocated, before the renderer changes, and with the viewport.
void Viewport::PinSumBuffers(bool pin)
{
    if (pin == mSumPinned || !mRenderer) return;
    for (Bitmap* b : { &mSum, &mSecondarySum })
    {
        if (!b->GetData()) continue;
        if (pin) (void)mRenderer->PinHostBuffer(b->GetData(), b->GetDataSize());
        else mRenderer->UnpinHostBuffer(b->GetData());
    }
    mSumPinned = pin;
}

void Viewport::SetSeed(uint64 seed)
{
    mRandomGenerator.Reset(seed);
    mHaltonSequence.GetRandom().Reset(seed ^ 0xA5A5A5A55A5A5A5AULL);
    mRngKey[0] = seed * 0x9E3779B97F4A7C15ULL + 1;
    mRngKey[1] = (seed ^ 0xD1B54A32D192ED03ULL) * 0xBF58476D1CE4E5B9ULL + 7;
}

bool Viewport::Resize(uint32 width, uint32 height)
{
    if (width > MAX_IMAGE_SIZE || height > MAX_IMAGE_SIZE || width == 0 || height == 0)
    {
        fprintf(stderr, "[rt] ERROR: Invalid viewport size\n");
        return false;
    }
    if (width == mWidth && height == mHeight) return true;
    mWidth = width; mHeight = height;
    PinSumBuffers(false);
    mSum.Init(width, height);
    mSecondarySum.Init(width, height);
    if (mRenderer && !mRenderer->Resize(width, height)) return false;
    // The reference fills mPixelSalt[width * height] from mRandomGenerator.GetVector4() here (Viewport.cpp:96-102).  Nothing reads the
    // salts, but the draws advance the generator the anti-aliasing offsets of every pass come from: they are part of the pass sequence.
    for (uint64 i = 0; i < (uint64)width * height; ++i) (void)mRandomGenerator.GetVector4();
    Reset();
    return true;
}

void Viewport::Reset()
{
    mHaltonSequence.Initialize(mParams.samplingParams.dimensions);
    ClearAccumulation();
}

// What Reset does to the accumulated frame, without re-drawing the Halton permutations: used when the renderer's film was cleared behind
// the viewport's back (a change of tile ownership), so that a sharded viewport draws the same per-pass constants as an unsharded one.
void Viewport::ClearAccumulation()
{
    mProgress = RenderingProgress();
    mSum.Clear();
    mSecondarySum.Clear();
    mSumDirty = false; mSecondarySumDirty = false;
    mCounters.Reset(); mTotalsAtLastPass.Reset(); mTotalsBeforeLastPass.Reset();
    mErrorEvaluatedAtPass = 0;
    BuildInitialBlocksList();   // Viewport::Reset -> BuildInitialBlocksList, Viewport.cpp:120-138
    if (mRenderer) { mRenderer->Reset(); mRenderer->SetActiveBlocks({}); }
}

bool Viewport::SetRenderer(const RendererPtr& renderer)
{
    PinSumBuffers(false);
    mRenderer = renderer;
    if (mRenderer && mWidth && mHeight) return mRenderer->Resize(mWidth, mHeight);
    return true;
}

bool Viewport::SetRenderingParams(const RenderingParams& params)
{
    if (params.maxRayDepth >= 255u || params.antiAliasingSpread < 0.0f) return false;
    if (params.samplingParams.dimensions > HaltonSequence::MaxDimensions) return false;
    mParams = params;
    return true;
}

bool Viewport::NextPassParams(const Camera& camera, RtPassParams& p)
{
    memset(&p, 0, sizeof(p));
    if (!camera.GetDesc(p.camera)) return false;

    mHaltonSequence.NextSample();
    const uint32 dims = mHaltonSequence.GetNumDimensions();
    mSeedStorage.resize(dims);
    for (uint32 i = 0; i < dims; ++i) mSeedStorage[i] = mHaltonSequence.GetInt(i);
    p.seed = mSeedStorage.data();
    p.numDimensions = dims;
    p.useBlueNoise = mParams.samplingParams.useBlueNoiseDithering ? 1u : 0u;

    // randomize pixel offset: normal-distributed, scaled by the anti-aliasing spread
    const Vector4 u = GetFloatNormal2(mRandomGenerator.GetFloat2());
    const Vector4 offset = u * mParams.antiAliasingSpread;
    p.sampleOffset[0] = offset.x; p.sampleOffset[1] = offset.y;

    p.passIndex = mProgress.passesFinished;
    p.maxRayDepth = mParams.maxRayDepth;
    p.minRussianRouletteDepth = mParams.minRussianRouletteDepth;
    p.lightSamplingStrategy = mParams.lightSamplingStrategy == LightSamplingStrategy::All ? RT_LIGHT_SAMPLING_ALL : RT_LIGHT_SAMPLING_SINGLE;
    Vector4 lw(1.0f), bw(1.0f);
    if (PathTracerMIS* pt = dynamic_cast<PathTracerMIS*>(mRenderer.get())) { lw = pt->mLightSamplingWeight; bw = pt->mBSDFSamplingWeight; }
    memcpy(p.lightSamplingWeight, &lw, 16);
    memcpy(p.bsdfSamplingWeight, &bw, 16);
    // a fresh key per pass so that per-pixel fallback streams do not repeat across passes
    p.rngKey[0] = mRngKey[0] + 0x9E3779B97F4A7C15ULL * (uint64)(mProgress.passesFinished + 1);
    p.rngKey[1] = mRngKey[1] ^ (0xC2B2AE3D27D4EB4FULL * (uint64)(mProgress.passesFinished + 1));
    return true;
}

bool Viewport::Render(const Camera& camera)
{
    if (mWidth == 0 || mHeight == 0) return false;
    if (!mRenderer)
    {
        fprintf(stderr, "[rt] ERROR: Viewport: Missing renderer\n");
        return false;
    }
    RtPassParams params;
    if (!NextPassParams(camera, params)) return false;
    const bool nothingLeft = mParams.adaptiveSettings.enable && mProgress.passesFinished > 0 && mBlocks.empty();
    if (!nothingLeft && !mRenderer->RenderPass(params)) return false;
    FinishPass();
    return true;
}

// the tail of Viewport::Render, Viewport.cpp:264-277
void Viewport::FinishPass()
{
    mProgress.passesFinished++;
    mSumDirty = true; mSecondarySumDirty = true;
    if (mProgress.passesFinished % 2 == 0 && mParams.adaptiveSettings.enable && mRenderer) UpdateBlocksList();
}

const RenderingProgress& Viewport::GetProgress()
{
    // Viewport::ComputeError (Viewport.cpp:179-183), lazily: the sum buffers must hold an even number of passes
    if (!mParams.adaptiveSettings.enable && mRenderer && mProgress.passesFinished > 0 && mProgress.passesFinished % 2 == 0 &&
        mErrorEvaluatedAtPass != mProgress.passesFinished)
    {
        std::vector<float> error;
        const RtBlock whole = { 0, mWidth, 0, mHeight };
        if (mRenderer->ComputeBlockErrors(mProgress.passesFinished, { whole }, error)) mProgress.averageError = error[0];
        mErrorEvaluatedAtPass = mProgress.passesFinished;
    }
    return mProgress;
}

void Viewport::BuildInitialBlocksList()   // Viewport.cpp:618-646
{
    mBlocks.clear();
    const uint32 blockSize = mParams.adaptiveSettings.maxBlockSize;
    if (mWidth == 0 || mHeight == 0 || blockSize == 0) { mProgress.activeBlocks = 0; return; }
    const uint32 rows = 1 + (mHeight - 1) / blockSize, columns = 1 + (mWidth - 1) / blockSize;
    for (uint32 j = 0; j < rows; ++j)
        for (uint32 i = 0; i < columns; ++i)
        {
            RtBlock block;
            block.minY = j * blockSize; block.maxY = std::min(mHeight, (j + 1) * blockSize);
            block.minX = i * blockSize; block.maxX = std::min(mWidth, (i + 1) * blockSize);
            mBlocks.push_back(block);
        }
    mProgress.activeBlocks = (uint32)mBlocks.size();
}

bool Viewport::UpdateBlocksList()   // Viewport.cpp:648-733
{
    const AdaptiveRenderingSettings& settings = mParams.adaptiveSettings;
    if (mProgress.passesFinished < settings.numInitialPasses) return true;
    std::vector<float> errors;
    if (!mRenderer->ComputeBlockErrors(mProgress.passesFinished, mBlocks, errors)) return false;
    ApplyBlockErrors(errors);
    if (mBlocks.empty()) return true;   // everything converged: Render() stops submitting passes (the reference renders zero tiles)
    return mRenderer->SetActiveBlocks(mBlocks);
}

// The list walk of Viewport::UpdateBlocksList given every block's ComputeBlockError (tests/golden/adaptive_kat.bin holds the reference's
// lists before and after, with the errors it computed).  Same walk as the reference: swap-and-pop removal, so a block swapped into slot i
// is not looked at in this update.  Every block that IS visited still sits at its original index, which is why the errors can be
// computed up front.
void Viewport::ApplyBlockErrors(std::vector<float>& errors)
{
    const AdaptiveRenderingSettings& settings = mParams.adaptiveSettings;
    if (mProgress.passesFinished < settings.numInitialPasses) return;
    std::vector<RtBlock> newBlocks;
    for (uint32 i = 0; i < mBlocks.size(); ++i)
    {
        const RtBlock block = mBlocks[i];
        const float blockError = errors[i];
        const uint32 width = block.maxX - block.minX, height = block.maxY - block.minY;
        if (blockError < settings.convergenceTreshold)
        {
            mBlocks[i] = mBlocks.back(); errors[i] = errors.back();   // block is fully converged - remove it
            mBlocks.pop_back(); errors.pop_back();
            continue;
        }
        if ((blockError < settings.subdivisionTreshold) && (width > settings.minBlockSize || height > settings.minBlockSize))
        {
            mBlocks[i] = mBlocks.back(); errors[i] = errors.back();   // block is somewhat converged - split it into two parts
            mBlocks.pop_back(); errors.pop_back();
            RtBlock childA = block, childB = block;
            if (width > height) { const uint32 halfPoint = (block.minX + block.maxX) / 2u; childA.maxX = halfPoint; childB.minX = halfPoint; }
            else { const uint32 halfPoint = (block.minY + block.maxY) / 2u; childA.maxY = halfPoint; childB.minY = halfPoint; }
            newBlocks.push_back(childA);
            newBlocks.push_back(childB);
        }
    }
    for (const RtBlock& block : newBlocks) mBlocks.push_back(block);
    mProgress.activePixels = 0;
    for (const RtBlock& block : mBlocks) mProgress.activePixels += (block.maxX - block.minX) * (block.maxY - block.minY);
    mProgress.converged = 1.0f - (float)mProgress.activePixels / (float)(mWidth * mHeight);
    mProgress.activeBlocks = (uint32)mBlocks.size();
}

// test hook (rth_viewport_kat_update_blocks): the pass counter and the errors come from a known-answer file instead of a renderer
void Viewport::UpdateBlocksListWithErrors(uint32 passesFinished, std::vector<float> errors)
{
    mProgress.passesFinished = passesFinished;
    ApplyBlockErrors(errors);
}

const Bitmap& Viewport::GetSumBuffer()
{
    if (mSumDirty && mRenderer)
    {
        PinSumBuffers(true);
        mRenderer->ReadSum(mSum.GetData(), nullptr);   // the secondary sum travels when somebody asks for it
        mSumDirty = false;
    }
    return mSum;
}

bool Viewport::SetPostprocessParams(const PostprocessParams& params)
{
    mPostprocessParams = params;
    return true;
}

const Bitmap& Viewport::GetFrontBuffer()
{
    Bitmap::InitData init;
    init.width = mWidth; init.height = mHeight; init.format = Bitmap::Format::B8G8R8A8_UNorm; init.linearSpace = false;
    if (mWidth && mHeight && (mFrontBuffer.GetWidth() != mWidth || mFrontBuffer.GetHeight() != mHeight || mFrontBuffer.GetFormat() != init.format)) mFrontBuffer.Init(init);
    if (mRenderer && mWidth && mHeight) mRenderer->PostProcess(mPostprocessParams, mProgress.passesFinished, reinterpret_cast<uint32*>(mFrontBuffer.GetBytes()));
    return mFrontBuffer;
}

const Bitmap& Viewport::GetSecondarySumBuffer()
{
    if (mSecondarySumDirty && mRenderer)
    {
        PinSumBuffers(true);
        mRenderer->ReadSum(nullptr, mSecondarySum.GetData());
        mSecondarySumDirty = false;
    }
    return mSecondarySum;
}

RayTracingCounters Viewport::GetTotalCounters()
{
    RayTracingCounters totals;
    if (mRenderer) mRenderer->GetCounters(totals);
    return totals;
}

const RayTracingCounters& Viewport::GetCounters()
{
    // The device keeps running totals; the per-pass figure of the reference is only exact when this is
    // polled after every pass.
    const RayTracingCounters totals = GetTotalCounters();
    mCounters = totals;
    mCounters.numRays -= mTotalsAtLastPass.numRays; mCounters.numShadowRays -= mTotalsAtLastPass.numShadowRays;
    mCounters.numShadowRaysHit -= mTotalsAtLastPass.numShadowRaysHit; mCounters.numPrimaryRays -= mTotalsAtLastPass.numPrimaryRays;
    mCounters.numRayBoxTests -= mTotalsAtLastPass.numRayBoxTests; mCounters.numPassedRayBoxTests -= mTotalsAtLastPass.numPassedRayBoxTests;
    mCounters.numRayTriangleTests -= mTotalsAtLastPass.numRayTriangleTests; mCounters.numPassedRayTriangleTests -= mTotalsAtLastPass.numPassedRayTriangleTests;
    mCounters.numMeshHits -= mTotalsAtLastPass.numMeshHits; mCounters.numAnalyticHits -= mTotalsAtLastPass.numAnalyticHits;
    mCounters.numShadowRayBoxTests -= mTotalsAtLastPass.numShadowRayBoxTests; mCounters.numShadowRayTriangleTests -= mTotalsAtLastPass.numShadowRayTriangleTests;
    mTotalsAtLastPass = totals;
    return mCounters;
}

// ---------------------------------------------------------------------------------------------------
// Renderer factory + the device renderer
// ---------------------------------------------------------------------------------------------------
IRenderer::~IRenderer() = default;

static int gRendererDevice = -1;
static std::vector<int> gRendererDevices;

void SetRendererDevice(int deviceIndex) { gRendererDevice = deviceIndex; gRendererDevices.clear(); }
int GetRendererDevice() { return gRendererDevice; }
void SetRendererDevices(const std::vector<int>& deviceIndices) { gRendererDevices = deviceIndices; }

// the devices of the next multi-device renderer: SetRendererDevices, else RTGPU_DEVICES ("0,1,2,3" / "all"); empty: one device
static bool MultiDeviceList(std::vector<int>& out, bool& all)
{
    out = gRendererDevices; all = false;
    if (out.empty())
        if (const char* e = getenv("RTGPU_DEVICES"))
        {
            if (strcmp(e, "all") == 0) { all = true; return true; }
            for (const char* p = e; *p;)
            {
                char* end = nullptr;
                const long d = strtol(p, &end, 10);
                if (end == p) break;
                out.push_back((int)d);
                p = (*end == ',') ? end + 1 : end;
                if (*end != ',' ) break;
            }
        }
    return out.size() > 1;
}

static int DefaultDevice()
{
    if (gRendererDevice >= 0) return gRendererDevice;
    if (const char* lr = getenv("LOCAL_RANK")) return atoi(lr);
    return 0;
}

RendererPtr CreateRenderer(const std::string& name, const Scene& scene)
{
    if (name == "Path Tracer MIS")
    {
        std::shared_ptr<PathTracerMIS> r(new PathTracerMIS(scene));
        if (!r->GetDeviceContext()) return nullptr;
        return r;
    }
    if (name == "Path Tracer")
    {
        std::shared_ptr<PathTracer> r(new PathTracer(scene));
        if (!r->GetDeviceContext()) return nullptr;
        return r;
    }
    if (name == "Debug")
    {
        std::shared_ptr<DebugRenderer> r(new DebugRenderer(scene));
        if (!r->GetDeviceContext()) return nullptr;
        return r;
    }
    if (name == "Light Tracer")
    {
        std::shared_ptr<LightTracer> r(new LightTracer(scene));
        if (!r->GetDeviceContext()) return nullptr;
        return r;
    }
    if (name == "VCM")
    {
        std::shared_ptr<VertexConnectionAndMerging> r(new VertexConnectionAndMerging(scene));
        if (!r->GetDeviceContext()) return nullptr;
        return r;
    }
    fprintf(stderr, "[rt] ERROR: unknown renderer '%s' (\"Path Tracer\", \"Path Tracer MIS\", \"Light Tracer\", \"VCM\", \"Debug\")\n", name.c_str());
    return nullptr;
}

static bool LoadBlueNoise(std::vector<uint16>& out)
{
    // search order: RT_DATA_DIR, ../Data (the reference's cwd-relative path, GenericSampler.cpp:13), the package data dir
    std::vector<std::string> candidates;
    if (const char* dir = getenv("RT_DATA_DIR")) candidates.push_back(std::string(dir) + "/BlueNoise128_RGBA16.dat");
    candidates.push_back("../Data/BlueNoise128_RGBA16.dat");
#ifdef RT_PACKAGE_DATA_DIR
    candidates.push_back(std::string(RT_PACKAGE_DATA_DIR) + "/BlueNoise128_RGBA16.dat");
#endif
    for (const std::string& path : candidates)
    {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) continue;
        out.resize(128 * 128 * 4);
        const bool ok = fread(out.data(), out.size() * sizeof(uint16), 1, f) == 1;
        fclose(f);
        if (ok) return true;
    }
    out.clear();
    fprintf(stderr, "[rt] ERROR: Failed to load blue noise texture\n");   // like the reference: dithering silently off
    return false;
}

PathTracerMIS::PathTracerMIS(const Scene& scene) : PathTracerMIS(scene, false) {}

PathTracerMIS::PathTracerMIS(const Scene& scene, bool wholeFrameOnOneDevice)
    : IRenderer(scene)
    , mLightSamplingWeight(1.0f)
    , mBSDFSamplingWeight(1.0f)
{
    std::vector<int> devices; bool all = false;
    const bool multi = !wholeFrameOnOneDevice && MultiDeviceList(devices, all);
    const int r = multi ? rtgpu_create_multi(all ? nullptr : devices.data(), all ? 0u : (uint32_t)devices.size(), &mCtx) : rtgpu_create(DefaultDevice(), &mCtx);
    if (r != RTGPU_OK)
    {
        fprintf(stderr, "[rt] ERROR: cannot create device context: %s\n", rtgpu_last_error());
        mCtx = nullptr;
    }
    LoadBlueNoise(mBlueNoise);
}

PathTracerMIS::~PathTracerMIS()
{
    if (mCtx) rtgpu_destroy(mCtx);
}

const char* PathTracerMIS::GetName() const { return "Path Tracer MIS"; }

bool PathTracerMIS::EnsureSceneUploaded()
{
    if (!mCtx) return false;
    if (mUploadedBuildId == mScene.GetBuildId()) return true;
    RtSceneDesc desc = mScene.GetDesc();
    desc.blueNoise = mBlueNoise.empty() ? nullptr : mBlueNoise.data();
    if (rtgpu_upload_scene(mCtx, &desc) != RTGPU_OK)
    {
        fprintf(stderr, "[rt] ERROR: scene upload failed: %s\n", rtgpu_last_error());
        return false;
    }
    mUploadedBuildId = mScene.GetBuildId();
    return true;
}

bool PathTracerMIS::Resize(uint32 width, uint32 height) { return mCtx && rtgpu_resize(mCtx, width, height) == RTGPU_OK; }
bool PathTracerMIS::Reset() { return mCtx && rtgpu_reset(mCtx) == RTGPU_OK; }
bool PathTracerMIS::SetShard(uint32 rank, uint32 worldSize) { RtgpuShard s = { rank, worldSize }; return mCtx && rtgpu_set_shard(mCtx, s) == RTGPU_OK; }

bool PathTracerMIS::RenderPass(const RtPassParams& params)
{
    if (!EnsureSceneUploaded()) return false;
    if (rtgpu_render_pass(mCtx, &params) != RTGPU_OK)
    {
        fprintf(stderr, "[rt] ERROR: render pass failed: %s\n", rtgpu_last_error());
        return false;
    }
    return true;
}

PathTracer::PathTracer(const Scene& scene) : PathTracerMIS(scene)
{
    if (GetDeviceContext() && rtgpu_set_integrator(GetDeviceContext(), RT_INTEGRATOR_PATH_TRACER, nullptr) != RTGPU_OK)
        fprintf(stderr, "[rt] ERROR: cannot select the plain path tracer: %s\n", rtgpu_last_error());
}
const char* PathTracer::GetName() const { return "Path Tracer"; }

LightTracer::LightTracer(const Scene& scene) : PathTracerMIS(scene, true)
{
    if (GetDeviceContext() && rtgpu_set_integrator(GetDeviceContext(), RT_INTEGRATOR_LIGHT_TRACER, nullptr) != RTGPU_OK)
        fprintf(stderr, "[rt] ERROR: cannot select the light tracer: %s\n", rtgpu_last_error());
}
const char* LightTracer::GetName() const { return "Light Tracer"; }

DebugRenderer::DebugRenderer(const Scene& scene) : PathTracerMIS(scene), mRenderingMode(DebugRenderingMode::TriangleID)
{
    if (GetDeviceContext() && rtgpu_set_integrator(GetDeviceContext(), RT_INTEGRATOR_DEBUG, nullptr) != RTGPU_OK)
        fprintf(stderr, "[rt] ERROR: cannot select the debug renderer: %s\n", rtgpu_last_error());
}
const char* DebugRenderer::GetName() const { return "Debug"; }
bool DebugRenderer::RenderPass(const RtPassParams& params)
{
    if (mAppliedMode != (int)mRenderingMode)
    {
        if (!GetDeviceContext() || rtgpu_set_debug_rendering_mode(GetDeviceContext(), (uint32_t)mRenderingMode) != RTGPU_OK) return false;
        mAppliedMode = (int)mRenderingMode;
    }
    return PathTracerMIS::RenderPass(params);
}

VertexConnectionAndMerging::VertexConnectionAndMerging(const Scene& scene)
    : PathTracerMIS(scene, true)
    , mVertexConnectingWeight(1.0f), mVertexMergingWeight(1.0f), mCameraConnectingWeight(1.0f)
    , mMaxPathLength(10), mInitialMergingRadius(0.02f), mMinMergingRadius(0.02f), mMergingRadiusMultiplier(1.0f)   // VertexConnectionAndMerging.cpp:53-71
    , mUseVertexConnection(true), mUseVertexMerging(true)
{
    memset(&mApplied, 0, sizeof(mApplied));
}

const char* VertexConnectionAndMerging::GetName() const { return "VCM"; }

bool VertexConnectionAndMerging::RenderPass(const RtPassParams& params)
{
    RtVcmParams vp; memset(&vp, 0, sizeof(vp));
    vp.maxPathLength = mMaxPathLength; vp.useVertexConnection = mUseVertexConnection ? 1u : 0u; vp.useVertexMerging = mUseVertexMerging ? 1u : 0u;
    vp.initialMergingRadius = mInitialMergingRadius; vp.minMergingRadius = mMinMergingRadius; vp.mergingRadiusMultiplier = mMergingRadiusMultiplier;
    memcpy(vp.bsdfSamplingWeight, &mBSDFSamplingWeight, 16); memcpy(vp.lightSamplingWeight, &mLightSamplingWeight, 16);
    memcpy(vp.vertexConnectingWeight, &mVertexConnectingWeight, 16); memcpy(vp.cameraConnectingWeight, &mCameraConnectingWeight, 16);
    memcpy(vp.vertexMergingWeight, &mVertexMergingWeight, 16);
    if (!mHaveApplied || memcmp(&vp, &mApplied, sizeof(vp)) != 0)
    {
        if (!GetDeviceContext() || rtgpu_set_integrator(GetDeviceContext(), RT_INTEGRATOR_VCM, &vp) != RTGPU_OK)
        {
            fprintf(stderr, "[rt] ERROR: cannot select the VCM integrator: %s\n", rtgpu_last_error());
            return false;
        }
        mApplied = vp; mHaveApplied = true;
    }
    return PathTracerMIS::RenderPass(params);
}

bool PathTracerMIS::PinHostBuffer(void* data, size_t bytes) { return mCtx && rtgpu_host_register(mCtx, data, bytes) == RTGPU_OK; }
void PathTracerMIS::UnpinHostBuffer(void* data) { if (mCtx) (void)rtgpu_host_unregister(mCtx, data); }
bool PathTracerMIS::ReadSum(float* sumRGB, float* secondaryRGB) { return mCtx && rtgpu_read_sum(mCtx, sumRGB, secondaryRGB) == RTGPU_OK; }

bool PathTracerMIS::ComputeBlockErrors(uint32 numPasses, const std::vector<RtBlock>& blocks, std::vector<float>& outErrors)
{
    outErrors.assign(blocks.size(), 0.0f);
    return mCtx && rtgpu_compute_block_errors(mCtx, numPasses, (uint32)blocks.size(), blocks.data(), outErrors.data()) == RTGPU_OK;
}

bool PathTracerMIS::SetActiveBlocks(const std::vector<RtBlock>& blocks)
{
    return mCtx && rtgpu_set_active_blocks(mCtx, (uint32)blocks.size(), blocks.data()) == RTGPU_OK;
}

bool PathTracerMIS::PostProcess(const PostprocessParams& params, uint32 numPasses, uint32* outBGRA)
{
    RtPostprocessParams p; memset(&p, 0, sizeof(p));
    memcpy(p.colorFilter, &params.colorFilter, 16);
    p.exposure = params.exposure; p.contrast = params.contrast; p.saturation = params.saturation;
    p.ditheringStrength = params.ditheringStrength; p.bloomFactor = params.bloomFactor;
    p.tonemapper = (uint32)params.tonemapper; p.numPasses = numPasses ? numPasses : 1u; p.ditherSeed = numPasses;
    if (!mCtx || rtgpu_postprocess(mCtx, &p, outBGRA) != RTGPU_OK)
    {
        fprintf(stderr, "[rt] ERROR: post-processing failed: %s\n", rtgpu_last_error());
        return false;
    }
    return true;
}

bool PathTracerMIS::GetCounters(RayTracingCounters& out)
{
    RtCounters c;
    if (!mCtx || rtgpu_get_counters(mCtx, &c) != RTGPU_OK) return false;
    out.numRays = c.numRays; out.numShadowRays = c.numShadowRays; out.numShadowRaysHit = c.numShadowRaysHit;
    out.numPrimaryRays = c.numPrimaryRays; out.numRayBoxTests = c.numRayBoxTests; out.numPassedRayBoxTests = c.numPassedRayBoxTests;
    out.numRayTriangleTests = c.numRayTriangleTests; out.numPassedRayTriangleTests = c.numPassedRayTriangleTests;
    out.numMeshHits = c.numMeshHits; out.numAnalyticHits = c.numAnalyticHits;
    out.numShadowRayBoxTests = c.numShadowRayBoxTests; out.numShadowRayTriangleTests = c.numShadowRayTriangleTests;
    return true;
}

} // namespace rt
