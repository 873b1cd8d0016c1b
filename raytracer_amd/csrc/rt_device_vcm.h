// rt_device_vcm.h -- device functions of the bidirectional integrator (reference: Core/Rendering/VertexConnectionAndMerging.cpp)
// beyond what the PathTracerMIS path already has: ILight::Emit, Illuminate / GetRadiance without solid-angle sampling and
// their emission pdfs, BSDF::Pdf, Camera::WorldToFilm / PdfW, the film splat, the packed photon fields, the photon hash
// grid query and Random::GetVector4.  Same arithmetic, same operation order as the reference's functions; every function
// cites the lines it follows.
#pragma once
#include "rt_device_core.h"

namespace rtd {

// ---- Random::GetIntVector4 / GetVector4, Core/Math/Random.cpp:83-126 (two 64-bit xorshift128+ lanes), per-pixel state ----
struct RandomSimd
{
    uint64_t seed0[2], seed1[2];   // mSeedSimd4[0], mSeedSimd4[1]
    __device__ __forceinline__ void resetPixel(uint32_t x, uint32_t y, uint64_t rngKey0, uint64_t rngKey1)
    {
        const uint64_t pix = (uint64_t)x | ((uint64_t)y << 32);
        seed0[0] = murmurFmix64(rngKey0 ^ pix ^ 0xA0761D6478BD642FULL);
        seed0[1] = murmurFmix64(rngKey1 ^ pix ^ 0xE7037ED1A0B428DBULL) | 1ULL;
        seed1[0] = murmurFmix64(rngKey0 + 0x8EBC6AF09C88C6E3ULL * (pix + 1));
        seed1[1] = murmurFmix64(rngKey1 + 0x589965CC75374CC3ULL * (pix + 1)) | 1ULL;
    }
    __device__ __forceinline__ V4 getVector4()
    {
        uint32_t i[4];
#pragma unroll
        for (int l = 0; l < 2; ++l)
        {
            const uint64_t s0 = seed1[l];
            uint64_t s1 = seed0[l];
            const uint64_t v = s0 + s1;
            s1 <<= 23;
            const uint64_t t0 = s0 >> 5;
            const uint64_t t1 = s1 >> 18;
            seed0[l] = s0;
            seed1[l] = (s0 ^ s1) ^ (t0 ^ t1);
            i[2 * l] = (uint32_t)v; i[2 * l + 1] = (uint32_t)(v >> 32);
        }
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = __uint_as_float((i[k] & 0x007fffffu) | 0x3f800000u) - 1.0f;   // :116-126
        return V4(f[0], f[1], f[2], f[3]);
    }
};

// ---- Camera::WorldToFilm / PdfW, Core/Scene/Camera.cpp:120-146 -------------------------------------------------------
RT_DEV bool cameraWorldToFilm(const RtCamera& cam, V4 worldPosition, V4& outFilmCoords)
{
    const V4 cameraSpacePosition = transformPoint(loadM4(cam.worldToScreen), worldPosition);
    if (cameraSpacePosition.z > 0.0f)
    {
        outFilmCoords = mulAdd(cameraSpacePosition / splat(cameraSpacePosition.w), splat(0.5f), splat(0.5f));   // BipolarToUnipolar
        return true;
    }
    return false;
}
RT_DEV float cameraDirectionPdfW(const RtCamera& cam, V4 direction)
{
    const float cosAtCamera = dot3(load4(cam.localToWorld + 8), direction);
    const float pdf = 0.25f / (Sqr(cam.tanHalfFoV) * (cosAtCamera * cosAtCamera * cosAtCamera) * cam.aspectRatio);
    return Max(0.0f, pdf);
}

// ---- Film::AccumulateColor(pos, color, random), Core/Rendering/Film.cpp:41-77: the receiving pixel (u = the GetVector4 draw)
RT_DEV bool filmSplatPixel(V4 pos, uint32_t width, uint32_t height, V4 u, uint32_t& outX, uint32_t& outY)
{
    const V4 filmSize((float)width, (float)height, 0.0f, 0.0f);
    const V4 filmCoords = pos * filmSize + V4(0.0f, 0.5f, 0.0f, 0.0f);
    int32_t ix = cvtRN(filmCoords.x), iy = cvtRN(filmCoords.y);
    {
        const float fracX = filmCoords.x - (float)ix, fracY = filmCoords.y - (float)iy;
        if (u.x < fracX) ix++;
        if (u.y < fracY) iy++;
    }
    const int32_t x = ix;
    const int32_t y = (int32_t)(height - 1u) - (int32_t)filmCoords.y;
    if ((uint32_t)x < width && (uint32_t)y < height) { outX = (uint32_t)x; outY = (uint32_t)y; return true; }
    return false;
}

// ---- Core/Math/Packed.h: PackedUnitVector3 (:15-61), PackedColorRgbHdr (:68-112) ------------------------------------
RT_DEV float changeSignIf(float v, bool flip) { return flip ? __uint_as_float(__float_as_uint(v) ^ 0x80000000u) : v; }
RT_DEV uint32_t packUnitVector(V4 input)
{
    const V4 vAbs = abs4(input);
    V4 n = input / splat(vAbs.x + vAbs.y + vAbs.z);
    if (input.z < 0.0f)
    {
        n = V4(n.y, n.x, n.y, n.x);
        const V4 t = splat(1.0f) - abs4(n);
        n = V4(changeSignIf(t.x, input.x < 0.0f), changeSignIf(t.y, input.y < 0.0f), changeSignIf(t.z, input.z < 0.0f), changeSignIf(t.w, input.w < 0.0f));
    }
    const int16_t u = (int16_t)cvtRN(n.x * 32767.0f), v = (int16_t)cvtRN(n.y * 32767.0f);
    return (uint32_t)(uint16_t)u | ((uint32_t)(uint16_t)v << 16);
}
RT_DEV V4 unpackUnitVector(uint32_t packed)
{
    const int16_t u = (int16_t)(packed & 0xFFFFu), v = (int16_t)(packed >> 16);
    V4 f = V4((float)u, (float)v, 0.0f, 0.0f) * (1.0f / 32767.0f);
    const V4 fAbs = abs4(f);
    f.z = 1.0f - fAbs.x - fAbs.y;
    const V4 t = max4(V4(0.0f - f.z, 0.0f - f.z, 0.0f - f.w, 0.0f - f.w), zero4());
    f = f + V4(changeSignIf(t.x, f.x > 0.0f), changeSignIf(t.y, f.y > 0.0f), changeSignIf(t.z, f.z > 0.0f), changeSignIf(t.w, f.w > 0.0f));
    return normalized3(f);
}
// the packed colour as {luminance float, co | cg << 16}
RT_DEV void packColorHdr(V4 color, float& outY, uint32_t& outChroma)
{
    const float ChromaScale = 16383.0f;
    V4 ycocg = splat(color.x) * V4(0.25f, 0.5f * ChromaScale, -0.25f * ChromaScale, 0.0f);
    ycocg = mulAdd(splat(color.y), V4(0.5f, 0.0f, 0.5f * ChromaScale, 0.0f), ycocg);
    ycocg = mulAdd(splat(color.z), V4(0.25f, -0.5f * ChromaScale, -0.25f * ChromaScale, 0.0f), ycocg);
    outY = ycocg.x;
    if (ycocg.x > 0.0f) ycocg = ycocg / splat(outY);
    const int16_t co = (int16_t)cvtRN(ycocg.y), cg = (int16_t)cvtRN(ycocg.z);
    outChroma = (uint32_t)(uint16_t)co | ((uint32_t)(uint16_t)cg << 16);
}
RT_DEV V4 unpackColorHdr(float y, uint32_t chroma)
{
    const int16_t co = (int16_t)(chroma & 0xFFFFu), cg = (int16_t)(chroma >> 16);
    const V4 cocg = V4((float)co, (float)cg, 0.0f, 0.0f) * (1.0f / 16383.0f);
    const float tmp = 1.0f - cocg.y;
    return max4(zero4(), V4(tmp + cocg.x, 1.0f + cocg.y, tmp - cocg.x, 0.0f) * y);
}

// ---- VertexConnectionAndMerging::Photon (.h:72-87), 32 bytes = two float4 ----------------------------------------------
struct Photon { float px, py, pz, lum; uint32_t chroma, direction; float dVM, dVCM; };

// ---- Core/Utils/HashGrid.h ---------------------------------------------------------------------------------------------
// `photons` is stored in mIndices order (photon j of the view = mPhotons[mIndices[j]]), so a cell's photons are contiguous
struct HashGridView
{
    const Photon* photons; const uint32_t* cellEnds;
    float boxMin[3]; float radiusSqr, invCellSize; uint32_t hashTableMask, numPhotons;
};
RT_DEV int32_t cvtT(float f) { return (f >= 2147483648.0f || f < -2147483648.0f || f != f) ? (int32_t)0x80000000 : (int32_t)f; }   // _mm_cvttps_epi32
RT_DEV uint32_t hashCellIndex(uint32_t x, uint32_t y, uint32_t z, uint32_t mask) { return ((x * 73856093u) ^ (y * 19349663u) ^ (z * 83492791u)) & mask; }   // :148-152
RT_DEV uint32_t hashCellOfPoint(const float boxMin[3], float invCellSize, uint32_t mask, float px, float py, float pz)   // :160-167
{
    const float cx = invCellSize * (px - boxMin[0]), cy = invCellSize * (py - boxMin[1]), cz = invCellSize * (pz - boxMin[2]);
    return hashCellIndex((uint32_t)cvtT(cx), (uint32_t)cvtT(cy), (uint32_t)cvtT(cz), mask);
}
// HashGrid::Process, :73-143: calls query(photonIndex) for every photon within the radius, cells and photons in the reference's order
template <typename Query>
RT_DEV void hashGridProcess(const HashGridView& g, V4 queryPos, Query& query)
{
    if (g.numPhotons == 0u) return;
    const V4 distMin = queryPos - V4(g.boxMin[0], g.boxMin[1], g.boxMin[2], 0.0f);
    const V4 cellCoords = mulSub(distMin, splat(g.invCellSize), splat(0.5f));
    const int32_t cx = cvtT(cellCoords.x), cy = cvtT(cellCoords.y), cz = cvtT(cellCoords.z);
    uint32_t numVisitedCells = 0, visitedCells[8];
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i)
    {
        const uint32_t x = (uint32_t)cx + (i & 1), y = (uint32_t)cy + ((i >> 1) & 1), z = (uint32_t)cz + (i >> 2);
        const uint32_t ci = hashCellIndex(x, y, z, g.hashTableMask);
        bool visited = false;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) if (j < numVisitedCells && visitedCells[j] == ci) visited = true;
        if (!visited) visitedCells[numVisitedCells++] = ci;
    }
    for (uint32_t i = 0; i < numVisitedCells; ++i)
    {
        const uint32_t ci = visitedCells[i];
        const uint32_t rangeStart = ci == 0 ? 0 : g.cellEnds[ci - 1], rangeEnd = g.cellEnds[ci];
        for (uint32_t j = rangeStart; j < rangeEnd; ++j)
        {
            const Photon& ph = g.photons[j];
            const float distSqr = sqrLength3(queryPos - V4(ph.px, ph.py, ph.pz, 0.0f));
            if (distSqr <= g.radiusSqr) query(j);
        }
    }
}

// ---- lights ------------------------------------------------------------------------------------------------------------
#define kSceneRadius (30.0f)   // BackgroundLight.cpp:16, DirectionalLight.cpp:14
RT_DEV float uniformSpherePdf() { return RTD_INV_PI / 4.0f; }                           // Geometry.h:22-25
RT_DEV float uniformCirclePdf(float radius) { return 1.0f / (RTD_PI * Sqr(radius)); }   // Geometry.h:32-35

// ILight::Illuminate with rendererSupportsSolidAngleSampling = false; also returns IlluminateResult::emissionPdfW
template <int kClass = 0>
RT_DEV V4 lightIlluminateBidir(const RtSceneDesc& d, const RtLight& L, const Intersection& isect, const float u[3], IlluminateResult& out, float& outEmissionPdfW)
{
    outEmissionPdfW = -1.0f;
    if (L.type == RT_LIGHT_AREA)   // AreaLight.cpp:79-104: a point on the surface; the normal goes through TransformPoint like the reference
    {
        out.directionToLight = zero4(); out.distance = -1.0f; out.directPdfW = -1.0f; out.cosAtLight = -1.0f;
        const M4 lightToWorld = loadM4(L.transform);
        V4 normalLocalSpace;
        const V4 samplePositionLocalSpace = shapeSampleArea(L.shapeKind, L.shapeParam, u, normalLocalSpace);
        const V4 lightPointWorldSpace = transformPoint(lightToWorld, samplePositionLocalSpace);
        const V4 normalWorldSpace = transformPoint(lightToWorld, normalLocalSpace);
        out.directionToLight = lightPointWorldSpace - isect.frame.r[3];
        const float sqrDistance = sqrLength3(out.directionToLight);
        out.distance = sqrtf(sqrDistance);
        out.directionToLight = out.directionToLight / out.distance;
        const float cosNormalDir = dot3(neg(normalWorldSpace), out.directionToLight);
        if (cosNormalDir < RTD_EPSILON) return zero4();
        const float invArea = 1.0f / shapeSurfaceArea(L.shapeKind, L.shapeParam);
        out.cosAtLight = cosNormalDir;
        out.directPdfW = invArea * sqrDistance / cosNormalDir;
        outEmissionPdfW = cosNormalDir * invArea * RTD_INV_PI;
        return load4(L.color);
    }
    const V4 radiance = lightIlluminate<kClass>(d, L, isect, u, out);
    switch (L.type)
    {
    case RT_LIGHT_BACKGROUND: outEmissionPdfW = uniformSpherePdf() * uniformCirclePdf(kSceneRadius); break;   // BackgroundLight.cpp:68
    case RT_LIGHT_DIRECTIONAL: outEmissionPdfW = out.directPdfW * uniformCirclePdf(kSceneRadius); break;      // DirectionalLight.cpp:85
    case RT_LIGHT_POINT: outEmissionPdfW = RTD_INV_PI / 4.0f; break;                                           // PointLight.cpp:41
    default: outEmissionPdfW = L.isDelta ? 1.0f : sphereCapPdf(L.cosAngle); break;                             // SpotLight.cpp:50
    }
    return radiance;
}

// ILight::GetRadiance with rendererSupportsSolidAngleSampling = false, plus *outEmissionPdfW
template <int kClass = 0>
RT_DEV V4 lightGetRadianceBidir(const RtSceneDesc& d, const RtLight& L, const Ray& lray, V4 hitPoint, float cosAtLight, float& outDirectPdfA, float& outEmissionPdfW)
{
    switch (L.type)
    {
    case RT_LIGHT_AREA:          // AreaLight.cpp:109-147
    {
        if (cosAtLight < RTD_EPSILON) return zero4();
        const float invArea = 1.0f / shapeSurfaceArea(L.shapeKind, L.shapeParam);
        outDirectPdfA = invArea;
        outEmissionPdfW = cosAtLight * invArea * RTD_INV_PI;
        return load4(L.color);
    }
    case RT_LIGHT_BACKGROUND:    // BackgroundLight.cpp:78-92
        outDirectPdfA = uniformHemispherePdf();
        outEmissionPdfW = uniformSpherePdf() * uniformCirclePdf(kSceneRadius);
        return backgroundColor<kClass>(d, L, lray.dir);
    case RT_LIGHT_DIRECTIONAL:   // DirectionalLight.cpp:94-121
        if (L.isDelta) return zero4();
        if (dot3(lray.dir, V4(0, 0, 1, 0)) > -L.cosAngle) return zero4();
        outDirectPdfA = sphereCapPdf(L.cosAngle);
        outEmissionPdfW = outDirectPdfA * uniformCirclePdf(kSceneRadius);
        return load4(L.color);
    default:
        return zero4();
    }
}

// ILight::Emit
struct EmitResult { V4 position, direction; float directPdfA, emissionPdfW, cosAtLight; };
template <int kClass = 0>
RT_DEV V4 lightEmit(const RtSceneDesc& d, const RtLight& L, const float up[3], const float ud[2], EmitResult& out)
{
    const M4 lightToWorld = loadM4(L.transform);
    switch (L.type)
    {
    case RT_LIGHT_AREA:          // AreaLight.cpp:149-185
    {
        V4 normalLocalSpace;
        const V4 samplePositionLocalSpace = shapeSampleArea(L.shapeKind, L.shapeParam, up, normalLocalSpace);
        out.position = transformPoint(lightToWorld, samplePositionLocalSpace);
        V4 tangentLocalSpace, bitangentLocalSpace;
        buildOrthonormalBasis(normalLocalSpace, tangentLocalSpace, bitangentLocalSpace);
        const V4 randomDir = getHemisphereCos(ud[0], ud[1]);
        const V4 dirLocalSpace = randomDir.x * tangentLocalSpace + randomDir.y * bitangentLocalSpace + randomDir.z * normalLocalSpace;
        out.direction = transformVector(lightToWorld, dirLocalSpace);
        const float cosAtLight = randomDir.z;
        const float invArea = 1.0f / shapeSurfaceArea(L.shapeKind, L.shapeParam);
        out.cosAtLight = cosAtLight;
        out.directPdfA = invArea;
        out.emissionPdfW = invArea * cosAtLight * RTD_INV_PI;
        return load4(L.color) * cosAtLight;
    }
    case RT_LIGHT_BACKGROUND:    // BackgroundLight.cpp:92-116
    {
        out.direction = getSphere(ud[0], ud[1]);
        const V4 uv = getCircle(up[0], up[1]);
        V4 u, v;
        buildOrthonormalBasis(out.direction, u, v);
        out.position = kSceneRadius * (u * uv.x + v * uv.y - out.direction);
        out.directPdfA = uniformHemispherePdf();
        out.emissionPdfW = uniformSpherePdf() * uniformCirclePdf(kSceneRadius);
        out.cosAtLight = 1.0f;
        return backgroundColor<kClass>(d, L, neg(out.direction));
    }
    case RT_LIGHT_DIRECTIONAL:   // DirectionalLight.cpp:120-135 (SampleDirection :47-78); the origin disc is NOT transformed
    {
        V4 dir = zero4();
        if (L.isDelta) { out.directPdfA = 1.0f; dir = V4(0, 0, 1, 0); }
        else
        {
            out.directPdfA = sphereCapPdf(L.cosAngle);
            const float phi = RTD_2PI * ud[1];
            const V4 sinCosPhi = sinCos(phi);
            float cosTheta = Lerp(L.cosAngle, 1.0f, ud[0]);
            float sinThetaSqr = 1.0f - Sqr(cosTheta);
            float sinTheta = sqrtf(sinThetaSqr);
            dir.x = sinTheta * sinCosPhi.x; dir.y = sinTheta * sinCosPhi.y; dir.z = cosTheta;
            dir = normalized3(dir);
        }
        out.direction = transformVector(lightToWorld, neg(dir));
        const V4 uv = getCircle(up[0], up[1]);
        out.position = V4(uv.x, uv.y, -1.0f, 0.0f) * kSceneRadius;
        out.cosAtLight = 1.0f;
        out.emissionPdfW = out.directPdfA * uniformCirclePdf(kSceneRadius);
        return load4(L.color);
    }
    case RT_LIGHT_POINT:         // PointLight.cpp:51-62
        out.position = lightToWorld.r[3];
        out.direction = getSphere(ud[0], ud[1]);
        out.emissionPdfW = RTD_INV_PI / 4.0f;
        out.directPdfA = 1.0f;
        out.cosAtLight = 1.0f;
        return load4(L.color);
    default:                     // RT_LIGHT_SPOT, SpotLight.cpp:63-93 (the direction stays in light space, like the reference)
    {
        if (L.isDelta) { out.emissionPdfW = 1.0f; out.direction = V4(0, 0, 1, 0); }
        else
        {
            const float phi = RTD_2PI * ud[1];
            const V4 sinCosPhi = sinCos(phi);
            float cosTheta = Lerp(L.cosAngle, 1.0f, ud[0]);
            float sinThetaSqr = 1.0f - Sqr(cosTheta);
            float sinTheta = sqrtf(sinThetaSqr);
            V4 dir = zero4();
            dir.x = sinTheta * sinCosPhi.x; dir.y = sinTheta * sinCosPhi.y; dir.z = cosTheta;
            out.direction = normalized3(dir);
            out.emissionPdfW = sphereCapPdf(L.cosAngle);
        }
        out.position = lightToWorld.r[3];
        out.directPdfA = 1.0f;
        out.cosAtLight = 1.0f;
        return load4(L.color);
    }
    }
}

RT_DEV bool bsdfIsDelta(uint32_t bsdf) { return bsdf == RT_BSDF_DIELECTRIC || bsdf == RT_BSDF_METAL; }   // BSDF::IsDelta
// BSDF::Pdf(ctx, dir) -- the pdf expressions of Evaluate, 0 where Evaluate returns black without writing them
RT_DEV float bsdfPdf(uint32_t bsdf, const RtMaterial& mat, const MatParams& mp, V4 outgoingDir, V4 incomingDir, bool reverse)
{
    float fwd = 0.0f, rev = 0.0f;
    bsdfEvaluate(bsdf, mat, mp, outgoingDir, incomingDir, fwd, &rev);
    return reverse ? rev : fwd;
}

} // namespace rtd
