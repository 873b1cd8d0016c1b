// oracle/rto_vcm.h -- TEST INFRASTRUCTURE (see rto_core.h): scalar CPU restatement of the reference's bidirectional
// integrator, Core/Rendering/VertexConnectionAndMerging.cpp ("VCM"), with the pieces it needs beyond the PathTracerMIS
// path: Camera::WorldToFilm / PdfW, the jittered film splat, the packed photon fields, HashGrid and Random::GetVector4.
//
// PARITY STATUS.  The building blocks (ILight::Emit, Illuminate/GetRadiance without solid-angle sampling, BSDF reverse
// pdfs, Camera::WorldToFilm/PdfW, Film::AccumulateColor(pos), PackedUnitVector3 / PackedColorRgbHdr) are pinned bit-exactly
// by golden vectors from the reference's own translation units (tests/golden/light_emit.kat ... packed_photon.kat).
// The renderer itself and Utils/HashGrid.h cannot be compiled here (both include Utils/Profiler.h -> <Windows.h>), and the
// reference's VCM image is not a function of the scene: light sub-paths and film jitter draw from the PER-THREAD generator
// (ctx.randomGenerator, seeded from entropy, Viewport.cpp:42) and photons are merged in thread order.  The integrator level
// is therefore "parity unpinned" against reference output; it is checked the way the reference checks it -- the furnace
// tests of Tests/RaytracingTests.cpp with their tolerances -- and against the PathTracerMIS oracle in expectation.
// Convention chosen here (and by the device path): every draw the reference takes from ctx.randomGenerator comes from a
// per-pixel generator (scalar xoroshiro128+ = Sampler::fallback, plus the two xorshift128+ lanes of GetVector4) seeded
// from RtPassParams::rngKey and the pixel; photons are ordered by pixel (row-major) then by path vertex.
#pragma once
#include "rto_core.h"
#include <vector>

namespace rto {

// ---- Random::GetIntVector4 / GetVector4, Core/Math/Random.cpp:83-126 (two 64-bit xorshift128+ lanes) ----------------
struct RandomSimd
{
    uint64_t seed0[2], seed1[2];   // mSeedSimd4[0], mSeedSimd4[1] as two uint64 lanes each
    void resetPixel(uint32_t x, uint32_t y, const uint64_t rngKey[2])
    {
        const uint64_t pix = (uint64_t)x | ((uint64_t)y << 32);
        seed0[0] = murmurFmix64(rngKey[0] ^ pix ^ 0xA0761D6478BD642FULL);
        seed0[1] = murmurFmix64(rngKey[1] ^ pix ^ 0xE7037ED1A0B428DBULL) | 1ULL;
        seed1[0] = murmurFmix64(rngKey[0] + 0x8EBC6AF09C88C6E3ULL * (pix + 1));
        seed1[1] = murmurFmix64(rngKey[1] + 0x589965CC75374CC3ULL * (pix + 1)) | 1ULL;
    }
    void nextInts(uint32_t out[4])
    {
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
            out[2 * l] = (uint32_t)v; out[2 * l + 1] = (uint32_t)(v >> 32);
        }
    }
    V4 getVector4()   // [0, 1) per lane: mantissa trick, :116-126
    {
        uint32_t i[4]; nextInts(i);
        float f[4];
        for (int k = 0; k < 4; ++k) { const uint32_t b = (i[k] & 0x007fffffu) | 0x3f800000u; memcpy(&f[k], &b, 4); f[k] -= 1.0f; }
        return V4(f[0], f[1], f[2], f[3]);
    }
};

// ---- Camera::WorldToFilm / PdfW, Core/Scene/Camera.cpp:120-146 -------------------------------------------------------
static inline bool cameraWorldToFilm(const RtCamera& cam, V4 worldPosition, V4& outFilmCoords)
{
    const V4 cameraSpacePosition = transformPoint(loadM4(cam.worldToScreen), worldPosition);
    if (cameraSpacePosition.z > 0.0f)
    {
        outFilmCoords = mulAdd(cameraSpacePosition / splat(cameraSpacePosition.w), splat(0.5f), splat(0.5f));   // BipolarToUnipolar
        return true;
    }
    return false;
}
static inline float cameraDirectionPdfW(const RtCamera& cam, V4 direction)
{
    const float cosAtCamera = dot3(load4(cam.localToWorld + 8), direction);
    const float pdf = 0.25f / (Sqr(cam.tanHalfFoV) * (cosAtCamera * cosAtCamera * cosAtCamera) * cam.aspectRatio);   // Cube(x) = x*x*x, Math.h
    return Max(0.0f, pdf);
}

// ---- Film::AccumulateColor(pos, color, random), Core/Rendering/Film.cpp:41-77: the receiving pixel -------------------
// returns false when the splat falls outside the film
// (u = the Random::GetVector4 draw of :51)
static inline bool filmSplatPixel(V4 pos, uint32_t width, uint32_t height, V4 u, uint32_t& outX, uint32_t& outY)
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
static inline float changeSignIf(float v, bool flip) { uint32_t b; memcpy(&b, &v, 4); if (flip) b ^= 0x80000000u; memcpy(&v, &b, 4); return v; }
static inline uint32_t packUnitVector(V4 input)
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
static inline V4 unpackUnitVector(uint32_t packed)
{
    const int16_t u = (int16_t)(packed & 0xFFFFu), v = (int16_t)(packed >> 16);
    V4 f = V4((float)u, (float)v, 0.0f, 0.0f) * (1.0f / 32767.0f);
    const V4 fAbs = abs4(f);
    f.z = 1.0f - fAbs.x - fAbs.y;
    const V4 t = max4(V4(0.0f - f.z, 0.0f - f.z, 0.0f - f.w, 0.0f - f.w), zero4());
    f = f + V4(changeSignIf(t.x, f.x > 0.0f), changeSignIf(t.y, f.y > 0.0f), changeSignIf(t.z, f.z > 0.0f), changeSignIf(t.w, f.w > 0.0f));
    return normalized3(f);
}
struct PackedColor { float y; int16_t co, cg; };
static inline PackedColor packColorHdr(V4 color)
{
    const float ChromaScale = 16383.0f;
    V4 ycocg = splat(color.x) * V4(0.25f, 0.5f * ChromaScale, -0.25f * ChromaScale, 0.0f);
    ycocg = mulAdd(splat(color.y), V4(0.5f, 0.0f, 0.5f * ChromaScale, 0.0f), ycocg);
    ycocg = mulAdd(splat(color.z), V4(0.25f, -0.5f * ChromaScale, -0.25f * ChromaScale, 0.0f), ycocg);
    PackedColor p; p.y = ycocg.x;
    if (ycocg.x > 0.0f) ycocg = ycocg / splat(p.y);
    p.co = (int16_t)cvtRN(ycocg.y); p.cg = (int16_t)cvtRN(ycocg.z);
    return p;
}
static inline V4 unpackColorHdr(PackedColor p)
{
    const V4 cocg = V4((float)p.co, (float)p.cg, 0.0f, 0.0f) * (1.0f / 16383.0f);
    const float tmp = 1.0f - cocg.y;
    return max4(zero4(), V4(tmp + cocg.x, 1.0f + cocg.y, tmp - cocg.x, 0.0f) * p.y);
}

// ---- VertexConnectionAndMerging::Photon (.h:72-87), 32 bytes -----------------------------------------------------------
struct Photon { float position[3]; PackedColor throughput; uint32_t direction; float dVM, dVCM; };
static_assert(sizeof(Photon) == 32, "Photon");

// ---- Core/Utils/HashGrid.h ---------------------------------------------------------------------------------------------
struct HashGrid
{
    V4 boxMin; std::vector<uint32_t> indices, cellEnds;
    float radiusSqr, cellSize, invCellSize; uint32_t hashTableMask;

    static uint32_t nextPowerOfTwo(uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++; return v; }   // Math.h:235-245
    static int32_t cvtT(float f) { return (f >= 2147483648.0f || f < -2147483648.0f || f != f) ? INT32_MIN : (int32_t)f; }                // _mm_cvttps_epi32
    uint32_t cellIndex(uint32_t x, uint32_t y, uint32_t z) const { return ((x * 73856093u) ^ (y * 19349663u) ^ (z * 83492791u)) & hashTableMask; }
    uint32_t cellIndex(const float* p) const   // :160-167
    {
        const V4 distMin = load3(p) - boxMin;
        const V4 coordF = invCellSize * distMin;
        return cellIndex((uint32_t)cvtT(coordF.x), (uint32_t)cvtT(coordF.y), (uint32_t)cvtT(coordF.z));
    }
    void build(const std::vector<Photon>& particles, float radius)   // :17-71
    {
        radiusSqr = Sqr(radius); cellSize = radius * 2.0f; invCellSize = 1.0f / cellSize;
        boxMin = splat(FLT_MAX);
        for (const Photon& p : particles) boxMin = min4(boxMin, load3(p.position));
        const uint32_t hashTableSize = nextPowerOfTwo((uint32_t)particles.size());
        hashTableMask = hashTableSize - 1;
        cellEnds.assign(hashTableSize, 0u);
        for (const Photon& p : particles) cellEnds[cellIndex(p.position)]++;
        uint32_t sum = 0;
        for (uint32_t& c : cellEnds) { const uint32_t temp = c; c = sum; sum += temp; }
        indices.resize(particles.size());
        for (uint32_t i = 0; i < (uint32_t)particles.size(); ++i) indices[cellEnds[cellIndex(particles[i].position)]++] = i;
    }
    template <typename Query>
    void process(V4 queryPos, const std::vector<Photon>& particles, Query& query) const   // :73-143
    {
        if (indices.empty()) return;
        const V4 distMin = queryPos - boxMin;
        const V4 cellCoords = mulSub(distMin, splat(invCellSize), splat(0.5f));
        const int32_t cx = cvtT(cellCoords.x), cy = cvtT(cellCoords.y), cz = cvtT(cellCoords.z);
        uint32_t numVisitedCells = 0, visitedCells[8];
        for (uint32_t i = 0; i < 8; ++i)
        {
            const uint32_t x = (uint32_t)cx + (i & 1), y = (uint32_t)cy + ((i >> 1) & 1), z = (uint32_t)cz + (i >> 2);
            const uint32_t ci = cellIndex(x, y, z);
            bool visited = false;
            for (uint32_t j = 0; j < numVisitedCells; ++j) if (visitedCells[j] == ci) { visited = true; break; }
            if (!visited) visitedCells[numVisitedCells++] = ci;
        }
        for (uint32_t i = 0; i < numVisitedCells; ++i)
        {
            const uint32_t ci = visitedCells[i];
            const uint32_t rangeStart = ci == 0 ? 0 : cellEnds[ci - 1], rangeEnd = cellEnds[ci];
            for (uint32_t j = rangeStart; j < rangeEnd; ++j)
            {
                const uint32_t particleIndex = indices[j];
                const float distSqr = sqrLength3(queryPos - load3(particles[particleIndex].position));
                if (distSqr <= radiusSqr) query(particleIndex);
            }
        }
    }
};

// =====================================================================================================
// VertexConnectionAndMerging
// =====================================================================================================
struct VcmSettings   // the public knobs, VertexConnectionAndMerging.h:35-53 with the constructor's defaults (.cpp:53-71)
{
    uint32_t maxPathLength = 10, useVertexConnection = 1, useVertexMerging = 1;
    float initialMergingRadius = 0.02f, minMergingRadius = 0.02f, mergingRadiusMultiplier = 1.0f;
    float bsdfSamplingWeight[4] = { 1, 1, 1, 1 }, lightSamplingWeight[4] = { 1, 1, 1, 1 }, vertexConnectingWeight[4] = { 1, 1, 1, 1 },
          cameraConnectingWeight[4] = { 1, 1, 1, 1 }, vertexMergingWeight[4] = { 1, 1, 1, 1 };
};

static const uint32_t kMaxLightVertices = 256;   // g_MaxLightVertices, .cpp:31
static inline float vcmMis(float pdf) { return pdf; }                                                        // :20-23
static inline float vcmPdfWtoA(float pdfW, float distance, float cosThere) { return pdfW * Abs(cosThere) / Sqr(distance); }   // :25-28

struct VcmPathState   // .h:98-112
{
    Ray ray; V4 throughput = splat(1.0f);
    float dVC = 0.0f, dVM = 0.0f, dVCM = 0.0f;
    uint32_t length = 1u; uint32_t lastSampledBsdfEvent = EV_NULL; bool lastSpecular = true, isFiniteLight = false;
};
struct VcmLightVertex { ShadingData shadingData; V4 throughput; float dVC, dVM, dVCM; uint8_t pathLength; };   // .h:59-70

struct VcmRenderer
{
    VcmSettings s;
    // PreRender state, .cpp:84-124
    float mergingRadiusVC = 0.0f, mergingRadiusVM = 0.0f;
    float vertexMergingNormalizationFactor = 0.0f;
    float misVertexMergingWeightFactorVC = 0.0f, misVertexConnectionWeightFactorVC = 0.0f;
    float misVertexMergingWeightFactorVM = 0.0f, misVertexConnectionWeightFactorVM = 0.0f;
    uint32_t lightPathsCount = 0;
    std::vector<Photon> photons;          // mPhotons: what the previous pass recorded
    std::vector<Photon> recorded;         // the per-thread lists, concatenated in pixel order
    HashGrid hashGrid;

    void preRender(uint32_t passNumber, uint32_t width, uint32_t height)
    {
        lightPathsCount = height * width;
        if (passNumber == 0) { mergingRadiusVC = s.initialMergingRadius; mergingRadiusVM = s.initialMergingRadius; recorded.clear(); }
        else
        {
            mergingRadiusVM = mergingRadiusVC;
            mergingRadiusVC *= s.mergingRadiusMultiplier;
            mergingRadiusVC = Max(mergingRadiusVC, s.minMergingRadius);
        }
        vertexMergingNormalizationFactor = 1.0f / (Sqr(mergingRadiusVM) * RTO_PI * lightPathsCount);
        {
            const float etaVCM = RTO_PI * Sqr(mergingRadiusVC) * lightPathsCount;
            misVertexMergingWeightFactorVC = (s.useVertexMerging && passNumber > 0) ? vcmMis(etaVCM) : 0.0f;
            misVertexConnectionWeightFactorVC = s.useVertexConnection ? vcmMis(1.f / etaVCM) : 0.0f;
        }
        {
            const float etaVCM = RTO_PI * Sqr(mergingRadiusVM) * lightPathsCount;
            misVertexMergingWeightFactorVM = s.useVertexMerging ? vcmMis(etaVCM) : 0.0f;
            misVertexConnectionWeightFactorVM = s.useVertexConnection ? vcmMis(1.f / etaVCM) : 0.0f;
        }
        // PreRender(ctx) + PreRenderGlobal(ctx) + PreRenderGlobal(), :126-170: last pass's photons become the merge set
        photons.swap(recorded);
        recorded.clear();
        if (s.useVertexMerging) hashGrid.build(photons, mergingRadiusVM);
    }
};

struct VcmCtx
{
    const RtSceneDesc* scene; const RtPassParams* params; VcmRenderer* r;
    Sampler sampler; RandomSimd simd; Counters* counters;
    uint32_t width, height; float* sum; float* secondarySum;   // film (float3 per pixel); secondarySum may be null
    uint32_t numLightVertices; VcmLightVertex lightVertices[kMaxLightVertices];
};

static inline bool vcmShadowed(VcmCtx& ctx, V4 origin, V4 dir, float distance)
{
    Hit hp; hp.objectId = RT_INVALID_OBJECT; hp.subObjectId = 0; hp.u = hp.v = 0.0f;
    hp.distance = distance * 0.999f;
    Ray shadowRay = makeRay(origin, dir);
    shadowRay.origin = shadowRay.origin + shadowRay.dir * 0.0001f;
    ctx.counters->c[C_SHADOW]++;
    if (sceneTraverseShadow(ctx.scene, shadowRay, hp, *ctx.counters)) return true;
    ctx.counters->c[C_SHADOW_HIT]++;
    return false;
}

// VertexConnectionAndMerging::AdvancePath, :493-578
static inline bool vcmAdvancePath(VcmCtx& ctx, VcmPathState& path, const ShadingData& sd, bool cameraPath)
{
    float sample[3];
    if (cameraPath) { sample[0] = ctx.sampler.getFloat(); sample[1] = ctx.sampler.getFloat(); sample[2] = ctx.sampler.getFloat(); }
    else { const V4 v = ctx.simd.getVector4(); sample[0] = v.x; sample[1] = v.y; sample[2] = v.z; }

    const RtMaterial& mat = ctx.scene->materials[sd.intersection.material];
    V4 incomingDirWorldSpace = zero4(); float bsdfDirPdf = 0.0f; uint32_t sampledEvent = EV_NULL;
    const V4 bsdfValue = materialSample(mat, sd, sample, incomingDirWorldSpace, bsdfDirPdf, sampledEvent);
    const float cosThetaOut = Abs(dot3(incomingDirWorldSpace, sd.intersection.frame.r[2]));
    if (sampledEvent == EV_NULL) return false;

    path.throughput = path.throughput * bsdfValue;
    if (almostZero4(path.throughput)) return false;

    path.ray = makeRay(sd.intersection.frame.r[3], incomingDirWorldSpace);
    path.ray.origin = path.ray.origin + path.ray.dir * 0.001f;
    path.lastSampledBsdfEvent = sampledEvent;
    path.length++;

    if (sampledEvent & EV_SPECULAR)
    {
        path.dVC *= vcmMis(cosThetaOut);
        path.dVM *= vcmMis(cosThetaOut);
        path.dVCM = 0.0f;
        path.lastSpecular = true;
    }
    else
    {
        const V4 outgoingLocal = worldToLocal(sd.intersection, sd.outgoingDirWorldSpace);
        const V4 incomingLocal = neg(worldToLocal(sd.intersection, incomingDirWorldSpace));
        const float bsdfRevPdf = bsdfPdf(mat.bsdf, mat, sd.mp, outgoingLocal, incomingLocal, true);
        const float invBsdfDirPdf = 1.0f / bsdfDirPdf;
        const float dVC = vcmMis(cosThetaOut * invBsdfDirPdf) * (path.dVC * vcmMis(bsdfRevPdf) + path.dVCM + ctx.r->misVertexMergingWeightFactorVC);
        const float dVM = vcmMis(cosThetaOut * invBsdfDirPdf) * (path.dVM * vcmMis(bsdfRevPdf) + path.dVCM * ctx.r->misVertexConnectionWeightFactorVC + 1.0f);
        path.dVC = dVC; path.dVM = dVM;
        path.dVCM = vcmMis(invBsdfDirPdf);
        path.lastSpecular = false;
    }
    return true;
}

// VertexConnectionAndMerging::GenerateLightSample, :428-491
static inline bool vcmGenerateLightSample(VcmCtx& ctx, VcmPathState& outPath)
{
    const uint32_t numLights = ctx.scene->numLights;
    if (numLights == 0) return false;
    const float lightPickProbability = 1.0f / (float)numLights;
    const uint32_t lightIndex = ctx.sampler.fallbackInt() % numLights;
    const RtLight& light = ctx.scene->lights[lightIndex];
    const V4 ps = ctx.simd.getVector4(); const V4 ds = ctx.simd.getVector4();
    const float up[3] = { ps.x, ps.y, ps.z }, ud[2] = { ds.x, ds.y };
    EmitResult er; er.position = zero4(); er.direction = zero4(); er.directPdfA = er.emissionPdfW = er.cosAtLight = 0.0f;
    const V4 throughput = lightEmit(ctx.scene, light, up, ud, er);
    if (almostZero4(throughput)) return false;
    er.directPdfA *= lightPickProbability;
    er.emissionPdfW *= lightPickProbability;
    const float emissionInvPdfW = 1.0f / er.emissionPdfW;
    er.position = er.position + er.direction * 0.0005f;
    outPath.ray = makeRay(er.position, er.direction);
    outPath.throughput = throughput * emissionInvPdfW;
    outPath.isFiniteLight = (light.flags & RT_LIGHT_FLAG_FINITE) != 0;
    {
        outPath.dVCM = vcmMis(er.directPdfA * emissionInvPdfW);
        if ((light.flags & RT_LIGHT_FLAG_DELTA) == 0)
        {
            const float cosAtLight = outPath.isFiniteLight ? er.cosAtLight : 1.0f;
            outPath.dVC = vcmMis(cosAtLight * emissionInvPdfW);
        }
        else outPath.dVC = 0.0f;
        outPath.dVM = outPath.dVC * ctx.r->misVertexConnectionWeightFactorVC;
    }
    return true;
}

// Film::AccumulateColor(pos, value, random) on the oracle's float3 buffers
static inline void vcmSplat(VcmCtx& ctx, V4 filmPos, V4 value, V4 jitter)
{
    uint32_t x, y;
    if (!filmSplatPixel(filmPos, ctx.width, ctx.height, jitter, x, y)) return;
    float* p = ctx.sum + 3 * ((size_t)y * ctx.width + x);
    p[0] = p[0] + value.x; p[1] = p[1] + value.y; p[2] = p[2] + value.z;
    if (ctx.secondarySum) { float* q = ctx.secondarySum + 3 * ((size_t)y * ctx.width + x); q[0] = q[0] + value.x; q[1] = q[1] + value.y; q[2] = q[2] + value.z; }
}

// VertexConnectionAndMerging::ConnectToCamera, :908-966
static inline void vcmConnectToCamera(VcmCtx& ctx, const VcmLightVertex& lv)
{
    const RtCamera& cam = ctx.params->camera;
    const V4 cameraPos = load4(cam.localToWorld + 12);
    const V4 samplePos = lv.shadingData.intersection.frame.r[3];
    V4 dirToCamera = cameraPos - samplePos;
    const float cameraDistanceSqr = sqrLength3(dirToCamera);
    const float cameraDistance = sqrtf(cameraDistanceSqr);
    dirToCamera = dirToCamera / cameraDistance;

    const RtMaterial& mat = ctx.scene->materials[lv.shadingData.intersection.material];
    float bsdfPdfW = 0.0f, bsdfRevPdfW = 0.0f;
    const V4 cameraFactor = materialEvaluate(mat, lv.shadingData, neg(dirToCamera), bsdfPdfW, &bsdfRevPdfW);
    if (almostZero4(cameraFactor)) return;

    V4 filmPos;
    if (!cameraWorldToFilm(cam, samplePos, filmPos)) return;
    // CONVENTION: the reference draws the film jitter inside Film::AccumulateColor, i.e. only for connections that turn out
    // visible (:940-965).  Its generator is per-thread and entropy-seeded, so the stream position carries no meaning; here
    // the draw is taken when the connection is set up, so that the per-pixel stream does not depend on a visibility
    // result (a wavefront implementation only learns it one kernel later).  The draws are i.i.d.: same estimator.
    const V4 jitter = ctx.simd.getVector4();
    if (vcmShadowed(ctx, samplePos, dirToCamera, cameraDistance)) return;

    const float cosToCamera = dot3(dirToCamera, lv.shadingData.intersection.frame.r[2]);
    if (cosToCamera <= FLT_EPSILON) return;

    const float cameraPdfW = cameraDirectionPdfW(cam, neg(dirToCamera));
    const float cameraPdfA = cameraPdfW * cosToCamera / cameraDistanceSqr;
    const float wLight = vcmMis(cameraPdfA) * (ctx.r->misVertexMergingWeightFactorVC + lv.dVCM + lv.dVC * vcmMis(bsdfRevPdfW));
    const float misWeight = 1.0f / (wLight + 1.0f);
    V4 contribution = (cameraFactor * lv.throughput) * (misWeight * cameraPdfA / (cosToCamera));
    contribution = contribution * load4(ctx.r->s.cameraConnectingWeight);
    vcmSplat(ctx, filmPos, contribution, jitter);
}

// VertexConnectionAndMerging::TraceLightPath, :320-426
static inline void vcmTraceLightPath(VcmCtx& ctx)
{
    ctx.numLightVertices = 0;
    VcmPathState pathState;
    if (!vcmGenerateLightSample(ctx, pathState)) return;
    const RtSceneDesc* scene = ctx.scene;
    Hit hitPoint; hitPoint.subObjectId = 0; hitPoint.u = hitPoint.v = 0.0f;
    for (;;)
    {
        hitPoint.objectId = RT_INVALID_OBJECT;
        hitPoint.distance = INFINITY;
        sceneTraverse(scene, pathState.ray, hitPoint, *ctx.counters);
        ctx.counters->c[C_RAYS]++;
        if (hitPoint.distance == INFINITY) break;
        if (hitPoint.subObjectId == RT_LIGHT_OBJECT) break;

        VcmLightVertex& vertex = ctx.lightVertices[ctx.numLightVertices];
        ShadingData& sd = vertex.shadingData;
        sd.intersection.material = RT_NO_MATERIAL;
        sceneEvaluateIntersection(scene, pathState.ray, hitPoint, sd.intersection, *ctx.counters);
        sd.outgoingDirWorldSpace = neg(pathState.ray.dir);
        const RtMaterial& mat = scene->materials[sd.intersection.material];
        materialEvaluateShadingData(scene, mat, sd);
        {
            if (pathState.length > 1 || pathState.isFiniteLight) pathState.dVCM *= vcmMis(Sqr(hitPoint.distance));
            const float cosTheta = dot3(pathState.ray.dir, sd.intersection.frame.r[2]);
            const float invMis = 1.0f / vcmMis(Abs(cosTheta));
            pathState.dVCM *= invMis; pathState.dVC *= invMis; pathState.dVM *= invMis;
        }
        if (!bsdfIsDelta(mat.bsdf))
        {
            if (ctx.r->s.useVertexConnection)
            {
                ctx.numLightVertices++;
                vertex.pathLength = (uint8_t)pathState.length;
                vertex.throughput = pathState.throughput;
                vertex.dVC = pathState.dVC; vertex.dVM = pathState.dVM; vertex.dVCM = pathState.dVCM;
                vcmConnectToCamera(ctx, vertex);
            }
            if (ctx.r->s.useVertexMerging)
            {
                Photon ph; memset(&ph, 0, sizeof(ph));
                ph.position[0] = sd.intersection.frame.r[3].x; ph.position[1] = sd.intersection.frame.r[3].y; ph.position[2] = sd.intersection.frame.r[3].z;
                ph.direction = packUnitVector(sd.outgoingDirWorldSpace);
                ph.throughput = packColorHdr(pathState.throughput);
                ph.dVM = pathState.dVM; ph.dVCM = pathState.dVCM;
                ctx.r->recorded.push_back(ph);
            }
        }
        if (pathState.length + 2 > ctx.r->s.maxPathLength) break;
        if (!vcmAdvancePath(ctx, pathState, sd, false)) break;
    }
}

// VertexConnectionAndMerging::EvaluateLight, :580-635 (isect == nullptr for global lights)
static inline V4 vcmEvaluateLight(VcmCtx& ctx, uint32_t iteration, const RtLight& light, const float* invTransform, const Intersection* isect, const VcmPathState& ps)
{
    const M4 worldToLight = loadM4(invTransform);
    const Ray lightSpaceRay = transformRayUnsafe(worldToLight, ps.ray);
    const float cosAtLight = isect ? -dot3(isect->frame.r[2], ps.ray.dir) : 1.0f;
    const V4 lightSpaceHitPoint = isect ? transformPoint(worldToLight, isect->frame.r[3]) : zero4();
    float directPdfA = 0.0f, emissionPdfW = 0.0f;
    V4 lightContribution = lightGetRadiance(ctx.scene, light, lightSpaceRay, lightSpaceHitPoint, cosAtLight, directPdfA, &emissionPdfW, false);
    if (almostZero4(lightContribution)) return zero4();
    if (ps.length > 1)
    {
        const bool useVertexMerging = ctx.r->s.useVertexMerging && iteration > 0;
        if (useVertexMerging && !ctx.r->s.useVertexConnection)
        {
            if (!ps.lastSpecular) return zero4();
        }
        else
        {
            const float wCamera = vcmMis(directPdfA) * ps.dVCM + vcmMis(emissionPdfW) * ps.dVC;
            const float misWeight = 1.0f / (1.0f + wCamera);
            lightContribution = lightContribution * misWeight;
        }
    }
    lightContribution = lightContribution * load4(ctx.r->s.bsdfSamplingWeight);
    return lightContribution;
}

// VertexConnectionAndMerging::SampleLight, :637-717
static inline V4 vcmSampleLight(VcmCtx& ctx, const RtLight& light, const ShadingData& sd, const VcmPathState& ps)
{
    float u[3]; u[0] = ctx.sampler.getFloat(); u[1] = ctx.sampler.getFloat(); u[2] = ctx.sampler.getFloat();
    IlluminateResult ir;
    const V4 radiance = lightIlluminate(ctx.scene, light, sd.intersection, u, ir, false);
    if (almostZero4(radiance)) return zero4();
    const RtMaterial& mat = ctx.scene->materials[sd.intersection.material];
    float bsdfPdfW = 0.0f, bsdfRevPdfW = 0.0f;
    const V4 bsdfFactor = materialEvaluate(mat, sd, neg(ir.directionToLight), bsdfPdfW, &bsdfRevPdfW);
    if (almostZero4(bsdfFactor)) return zero4();
    if (vcmShadowed(ctx, sd.intersection.frame.r[3], ir.directionToLight, ir.distance)) return zero4();
    const float lightPickProbability = 1.0f;
    const bool isDeltaLight = (light.flags & RT_LIGHT_FLAG_DELTA) != 0;
    const float continuationProbability = 1.0f;
    bsdfPdfW *= isDeltaLight ? 0.0f : continuationProbability;
    bsdfRevPdfW *= continuationProbability;
    const float cosToLight = dot3(sd.intersection.frame.r[2], ir.directionToLight);
    if (cosToLight <= FLT_EPSILON) return zero4();
    const float wLight = vcmMis(bsdfPdfW / (lightPickProbability * ir.directPdfW));
    const float wCamera = vcmMis(ir.emissionPdfW * cosToLight / (ir.directPdfW * ir.cosAtLight)) * (ctx.r->misVertexMergingWeightFactorVC + ps.dVCM + ps.dVC * vcmMis(bsdfRevPdfW));
    const float misWeight = 1.0f / (wLight + 1.0f + wCamera);
    return (radiance * bsdfFactor) * (misWeight / (lightPickProbability * ir.directPdfW));
}

// VertexConnectionAndMerging::ConnectVertices, :746-821
static inline V4 vcmConnectVertices(VcmCtx& ctx, const VcmPathState& cameraPathState, const ShadingData& sd, const VcmLightVertex& lv)
{
    V4 lightDir = lv.shadingData.intersection.frame.r[3] - sd.intersection.frame.r[3];
    const float distanceSqr = sqrLength3(lightDir);
    const float distance = sqrtf(distanceSqr);
    lightDir = lightDir / distance;
    const float cosCameraVertex = dot3(sd.intersection.frame.r[2], lightDir);
    const float cosLightVertex = dot3(lv.shadingData.intersection.frame.r[2], neg(lightDir));
    if (cosCameraVertex <= 0.0f || cosLightVertex <= 0.0f) return zero4();
    const float geometryTerm = 1.0f / distanceSqr;

    float cameraBsdfPdfW = 0.0f, cameraBsdfRevPdfW = 0.0f;
    const V4 cameraFactor = materialEvaluate(ctx.scene->materials[sd.intersection.material], sd, neg(lightDir), cameraBsdfPdfW, &cameraBsdfRevPdfW);
    if (almostZero4(cameraFactor)) return zero4();
    float lightBsdfPdfW = 0.0f, lightBsdfRevPdfW = 0.0f;
    const V4 lightFactor = materialEvaluate(ctx.scene->materials[lv.shadingData.intersection.material], lv.shadingData, lightDir, lightBsdfPdfW, &lightBsdfRevPdfW);
    if (almostZero4(lightFactor)) return zero4();
    if (vcmShadowed(ctx, sd.intersection.frame.r[3], lightDir, distance)) return zero4();

    const float continuationProbability = 1.0f;
    lightBsdfPdfW *= continuationProbability;
    lightBsdfRevPdfW *= continuationProbability;
    const float cameraBsdfPdfA = vcmPdfWtoA(cameraBsdfPdfW, distance, cosLightVertex);
    const float lightBsdfPdfA = vcmPdfWtoA(lightBsdfPdfW, distance, cosCameraVertex);
    const float wLight = vcmMis(cameraBsdfPdfA) * (ctx.r->misVertexMergingWeightFactorVC + lv.dVCM + lv.dVC * vcmMis(lightBsdfRevPdfW));
    const float wCamera = vcmMis(lightBsdfPdfA) * (ctx.r->misVertexMergingWeightFactorVC + cameraPathState.dVCM + cameraPathState.dVC * vcmMis(cameraBsdfRevPdfW));
    const float misWeight = 1.0f / (wLight + 1.0f + wCamera);
    return (cameraFactor * lightFactor) * (geometryTerm * misWeight);
}

// VertexConnectionAndMerging::MergeVertices, :823-906
static inline V4 vcmMergeVertices(VcmCtx& ctx, const VcmPathState& cameraPathState, const ShadingData& sd)
{
    struct RangeQuery
    {
        VcmCtx& ctx; const VcmPathState& cps; const ShadingData& sd; V4 contribution;
        void operator()(uint32_t photonIndex)
        {
            const Photon& photon = ctx.r->photons[photonIndex];
            const V4 lightDirection = unpackUnitVector(photon.direction);
            const float cosToLight = dot3(sd.intersection.frame.r[2], lightDirection);
            if (cosToLight < FLT_EPSILON) return;
            float cameraBsdfDirPdfW = 0.0f, cameraBsdfRevPdfW = 0.0f;
            const V4 cameraBsdfFactor = materialEvaluate(ctx.scene->materials[sd.intersection.material], sd, neg(lightDirection), cameraBsdfDirPdfW, &cameraBsdfRevPdfW);
            if (almostZero4(cameraBsdfFactor)) return;
            const V4 throughput = unpackColorHdr(photon.throughput);
            const float wLight = photon.dVCM * ctx.r->misVertexConnectionWeightFactorVM + photon.dVM * vcmMis(cameraBsdfDirPdfW);
            const float wCamera = cps.dVCM * ctx.r->misVertexConnectionWeightFactorVM + cps.dVM * vcmMis(cameraBsdfRevPdfW);
            const float misWeight = 1.0f / (wLight + 1.0f + wCamera);
            const float weight = misWeight / cosToLight;
            contribution = mulAdd(cameraBsdfFactor * throughput, weight, contribution);
        }
    };
    RangeQuery query = { ctx, cameraPathState, sd, zero4() };
    ctx.r->hashGrid.process(sd.intersection.frame.r[3], ctx.r->photons, query);
    return query.contribution;
}

// VertexConnectionAndMerging::RenderPixel, :172-318
static inline V4 vcmRenderPixel(VcmCtx& ctx, const Ray& primaryRay, uint32_t iteration)
{
    vcmTraceLightPath(ctx);

    const RtSceneDesc* scene = ctx.scene;
    const VcmSettings& s = ctx.r->s;
    V4 resultColor = zero4();
    VcmPathState pathState; pathState.ray = primaryRay;
    {
        const float cameraPdf = cameraDirectionPdfW(ctx.params->camera, primaryRay.dir);
        pathState.dVC = 0.0f; pathState.dVM = 0.0f;
        pathState.dVCM = vcmMis(1.0f / cameraPdf);
        pathState.lastSpecular = true;
    }
    Hit hitPoint; hitPoint.subObjectId = 0; hitPoint.u = hitPoint.v = 0.0f;
    ShadingData shadingData; shadingData.intersection.material = RT_NO_MATERIAL;

    for (;;)
    {
        hitPoint.objectId = RT_INVALID_OBJECT;
        hitPoint.distance = INFINITY;
        sceneTraverse(scene, pathState.ray, hitPoint, *ctx.counters);
        ctx.counters->c[C_RAYS]++;

        if (hitPoint.distance == INFINITY)
        {
            V4 result = zero4();   // EvaluateGlobalLights, :733-744
            for (uint32_t g = 0; g < scene->numGlobalLights; ++g)
            {
                const RtLight& light = scene->lights[scene->globalLights[g]];
                result = result + vcmEvaluateLight(ctx, iteration, light, light.invTransform, nullptr, pathState);
            }
            resultColor = mulAdd(pathState.throughput, result, resultColor);
            break;
        }
        sceneEvaluateIntersection(scene, pathState.ray, hitPoint, shadingData.intersection, *ctx.counters);
        {
            const float cosTheta = dot3(pathState.ray.dir, shadingData.intersection.frame.r[2]);
            const float invMis = 1.0f / vcmMis(Abs(cosTheta));
            pathState.dVCM *= vcmMis(Sqr(hitPoint.distance));
            pathState.dVCM *= invMis; pathState.dVC *= invMis; pathState.dVM *= invMis;
        }
        if (hitPoint.subObjectId == RT_LIGHT_OBJECT)
        {
            const RtObject& obj = scene->objects[hitPoint.objectId];
            const V4 lightColor = vcmEvaluateLight(ctx, iteration, scene->lights[obj.lightIndex], obj.invTransform, &shadingData.intersection, pathState);
            resultColor = mulAdd(pathState.throughput, lightColor, resultColor);
            break;
        }
        shadingData.outgoingDirWorldSpace = neg(pathState.ray.dir);
        const RtMaterial& mat = scene->materials[shadingData.intersection.material];
        materialEvaluateShadingData(scene, mat, shadingData);
        resultColor = mulAdd(pathState.throughput, shadingData.mp.emission, resultColor);

        if (pathState.length >= s.maxPathLength) break;
        const bool isDeltaBsdf = bsdfIsDelta(mat.bsdf);

        if (!isDeltaBsdf && s.useVertexConnection)   // SampleLights, :719-731
        {
            V4 accumulatedColor = zero4();
            for (uint32_t i = 0; i < scene->numLights; ++i) accumulatedColor = accumulatedColor + vcmSampleLight(ctx, scene->lights[i], shadingData, pathState);
            accumulatedColor = accumulatedColor * load4(s.lightSamplingWeight);
            resultColor = mulAdd(pathState.throughput, accumulatedColor, resultColor);
        }
        const uint32_t numLightVertices = ctx.numLightVertices;
        if (!isDeltaBsdf && s.useVertexConnection && numLightVertices > 0)
        {
            V4 vertexConnectionColor = zero4();
            for (uint32_t i = 0; i < numLightVertices; ++i)
            {
                const VcmLightVertex& lv = ctx.lightVertices[i];
                if (lv.pathLength + pathState.length + 1u > s.maxPathLength) break;
                vertexConnectionColor = mulAdd(lv.throughput, vcmConnectVertices(ctx, pathState, shadingData, lv), vertexConnectionColor);
            }
            vertexConnectionColor = vertexConnectionColor * load4(s.vertexConnectingWeight);
            resultColor = mulAdd(pathState.throughput, vertexConnectionColor, resultColor);
        }
        if (!isDeltaBsdf && s.useVertexMerging && iteration > 0)
        {
            V4 vertexMergingColor = vcmMergeVertices(ctx, pathState, shadingData);
            vertexMergingColor = vertexMergingColor * load4(s.vertexMergingWeight);
            resultColor = mulAdd(pathState.throughput * vertexMergingColor, ctx.r->vertexMergingNormalizationFactor, resultColor);
        }
        if (pathState.length > s.maxPathLength) break;
        if (!vcmAdvancePath(ctx, pathState, shadingData, true)) break;
    }
    return resultColor;
}

// =====================================================================================================
// LightTracer::RenderPixel, Core/Rendering/LightTracer.cpp:25-183 (renderer "Light Tracer"): one light path per pixel, every vertex
// connected to the camera; the pixel's own colour is zero.  Same generator convention as VCM (per-pixel streams, splat jitter drawn
// when the connection is set up).  maxRayDepth comes from the rendering parameters.
// =====================================================================================================
static inline void lightTracerPixel(VcmCtx& ctx)
{
    const RtSceneDesc* scene = ctx.scene;
    uint32_t depth = 0;
    if (scene->numLights == 0) return;
    const float lightPickingProbability = 1.0f / (float)scene->numLights;
    const uint32_t lightIndex = ctx.sampler.fallbackInt() % scene->numLights;
    const RtLight& light = scene->lights[lightIndex];
    const V4 ps = ctx.simd.getVector4(); const V4 ds = ctx.simd.getVector4();
    const float up[3] = { ps.x, ps.y, ps.z }, ud[2] = { ds.x, ds.y };
    EmitResult er; er.position = zero4(); er.direction = zero4(); er.directPdfA = er.emissionPdfW = er.cosAtLight = 0.0f;
    V4 throughput = lightEmit(scene, light, up, ud, er);
    if (almostZero4(throughput)) return;
    er.emissionPdfW *= lightPickingProbability;
    er.position = er.position + er.direction * 0.0005f;
    Ray ray = makeRay(er.position, er.direction);
    throughput = throughput * (1.0f / er.emissionPdfW);
    Hit hitPoint; hitPoint.subObjectId = 0; hitPoint.u = hitPoint.v = 0.0f;
    ShadingData sd; sd.intersection.material = RT_NO_MATERIAL;
    const RtCamera& cam = ctx.params->camera;
    for (;;)
    {
        hitPoint.objectId = RT_INVALID_OBJECT;
        hitPoint.distance = INFINITY;
        sceneTraverse(scene, ray, hitPoint, *ctx.counters);
        if (hitPoint.distance == INFINITY) break;
        if (hitPoint.subObjectId == RT_LIGHT_OBJECT) break;
        if (hitPoint.distance < FLT_MAX)
        {
            sceneEvaluateIntersection(scene, ray, hitPoint, sd.intersection, *ctx.counters);
            sd.outgoingDirWorldSpace = neg(ray.dir);
            materialEvaluateShadingData(scene, scene->materials[sd.intersection.material], sd);
        }
        if (depth >= ctx.params->maxRayDepth) break;
        const RtMaterial& mat = scene->materials[sd.intersection.material];
        {   // connect to camera, :113-153
            const V4 cameraPos = load4(cam.localToWorld + 12);
            const V4 samplePos = sd.intersection.frame.r[3];
            V4 dirToCamera = cameraPos - samplePos;
            const float cameraDistanceSqr = sqrLength3(dirToCamera);
            const float cameraDistance = sqrtf(cameraDistanceSqr);
            dirToCamera = dirToCamera / cameraDistance;
            float bsdfPdfW = 0.0f;
            const V4 cameraFactor = materialEvaluate(mat, sd, neg(dirToCamera), bsdfPdfW);
            if (!almostZero4(cameraFactor))
            {
                V4 filmPos;
                if (cameraWorldToFilm(cam, samplePos, filmPos))
                {
                    const V4 jitter = ctx.simd.getVector4();   // convention: drawn at set-up (the reference draws it inside AccumulateColor)
                    Hit hp; hp.objectId = RT_INVALID_OBJECT; hp.subObjectId = 0; hp.u = hp.v = 0.0f;
                    hp.distance = cameraDistance * 0.999f;
                    const Ray shadowRay = makeRay(samplePos + sd.intersection.frame.r[2] * 0.0001f, dirToCamera);   // offset along the NORMAL, :138
                    ctx.counters->c[C_SHADOW]++;
                    if (!sceneTraverseShadow(scene, shadowRay, hp, *ctx.counters))
                    {
                        ctx.counters->c[C_SHADOW_HIT]++;
                        const float cameraPdfA = cameraDirectionPdfW(cam, neg(dirToCamera)) / cameraDistanceSqr;
                        const V4 contribution = (cameraFactor * throughput) * cameraPdfA;
                        vcmSplat(ctx, filmPos, contribution, jitter);
                    }
                }
            }
        }
        const V4 sv = ctx.simd.getVector4();
        const float sample[3] = { sv.x, sv.y, sv.z };
        V4 incomingDirWorldSpace = zero4(); float pdf = 0.0f; uint32_t event = EV_NULL;
        const V4 bsdfValue = materialSample(mat, sd, sample, incomingDirWorldSpace, pdf, event);
        throughput = throughput * bsdfValue;
        if (almostZero4(throughput)) break;
        ray = makeRay(sd.intersection.frame.r[3], incomingDirWorldSpace);
        ray.origin = ray.origin + ray.dir * 0.001f;
        depth++;
    }
    ctx.counters->c[C_RAYS] += (uint64_t)depth + 1;
}

} // namespace rto
