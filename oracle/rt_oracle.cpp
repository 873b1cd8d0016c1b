// oracle/rt_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C entry points of the CPU oracle (a scalar restatement of the reference's PathTracerMIS pass, see
// rto_core.h / rto_math.h for the per-function file:line citations).
//
// Pinning status: the FULL reference renderer cannot be built in this image without stand-in headers
// (Core/Scene/Scene.cpp and Core/Rendering/Renderer.cpp include <Windows.h> through
// Core/Utils/Profiler.h:11; Core/Utils/Memory.cpp:8 includes it directly), so the oracle is pinned
//   (a) function by function against golden vectors produced by the reference's OWN translation units
//       that do compile unmodified (oracle/ref_harness -> oracle/_ref/, fixtures in tests/golden/), and
//   (b) end to end against the reference's own six RenderingTest furnace cases
//       (Tests/RaytracingTests.cpp:263-523), restated in tests/test_furnace_oracle.py.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include "rto_core.h"
#include "rto_vcm.h"

#include <stdlib.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <thread>
#include <atomic>

using namespace rto;

extern "C" {

// Viewport::RenderTile pixel loop (Core/Rendering/Viewport.cpp:305-357) over rows [y0, y1)
static void renderRows(const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height,
                       uint32_t shardRank, uint32_t shardWorld, uint32_t y0, uint32_t y1,
                       float* sum, float* secondary, Counters* counters, int integrator = 0)
{
    RenderCtx ctx;
    ctx.scene = scene; ctx.params = params; ctx.counters = counters;
    ctx.lightSamplingWeight = load4(params->lightSamplingWeight);
    ctx.bsdfSamplingWeight = load4(params->bsdfSamplingWeight);
    ctx.sampler.seed = params->seed;
    ctx.sampler.numDims = params->numDimensions;
    ctx.sampler.blueNoise = scene->blueNoise;
    ctx.sampler.blueNoiseLayers = (scene->blueNoise && params->useBlueNoise) ? 4u : 0u;   // GenericSampler.cpp:69-73

    // filmSize = FromIntegers(w, h, 1, 1); invSize = VECTOR_ONE2 / filmSize   (Viewport.cpp:300-301)
    const V4 invSize(1.0f / (float)(int32_t)width, 1.0f / (float)(int32_t)height, 0.0f / 1.0f, 0.0f / 1.0f);
    const V4 sampleOffset(params->sampleOffset[0], params->sampleOffset[1], 0.0f, 0.0f);
    const bool evenPass = (params->passIndex % 2u) == 0u;
    const uint32_t tilesX = (width + 63u) / 64u;

    for (uint32_t y = y0; y < y1; ++y)
    {
        const uint32_t realY = height - 1u - y;
        for (uint32_t x = 0; x < width; ++x)
        {
            if (shardWorld > 1)
            {
                const uint32_t tile = (y / 64u) * tilesX + (x / 64u);
                if (tile % shardWorld != shardRank) continue;
            }
            const V4 coords = (V4((float)(int32_t)x, (float)(int32_t)realY, 0.0f, 0.0f) + sampleOffset) * invSize;
            ctx.sampler.resetPixel(x, y, params->rngKey);
            const Ray ray = cameraGenerateRay(params->camera, coords, ctx.sampler);
            const V4 color = integrator == 0 ? renderPixel(ctx, ray) : (integrator == 1 ? renderPixelPlain(ctx, ray) : renderPixelDebug(ctx, ray, (uint32_t)(integrator - 2)));
            float* px = sum + 3 * ((size_t)y * width + x);                 // Film::AccumulateColor Film.cpp:25-39
            px[0] = px[0] + color.x; px[1] = px[1] + color.y; px[2] = px[2] + color.z;
            if (evenPass && secondary)
            {
                float* sx = secondary + 3 * ((size_t)y * width + x);
                sx[0] = sx[0] + color.x; sx[1] = sx[1] + color.y; sx[2] = sx[2] + color.z;
            }
            counters->c[C_PRIMARY]++;
        }
    }
}

// One pass of the hot path on the CPU.  counters: uint64[16] accumulated (layout of RtCounters).
// numThreads <= 1: single thread.  Rows are split statically; the result does not depend on numThreads.
static int renderPassImpl(const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height,
                          uint32_t shardRank, uint32_t shardWorld, float* sum, float* secondary, uint64_t* counters, int numThreads, int integrator);
int rto_render_pass(const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height,
                    uint32_t shardRank, uint32_t shardWorld, float* sum, float* secondary, uint64_t* counters, int numThreads)
{
    return renderPassImpl(scene, params, width, height, shardRank, shardWorld, sum, secondary, counters, numThreads, 0);
}
// the renderer "Path Tracer" (Core/Rendering/PathTracer.cpp)
int rto_render_pass_plain(const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height,
                          uint32_t shardRank, uint32_t shardWorld, float* sum, float* secondary, uint64_t* counters, int numThreads)
{
    return renderPassImpl(scene, params, width, height, shardRank, shardWorld, sum, secondary, counters, numThreads, 1);
}
// the renderer "Debug" (Core/Rendering/DebugRenderer.cpp) in DebugRenderingMode `mode`
int rto_render_pass_debug(const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height, uint32_t mode,
                          float* sum, float* secondary, uint64_t* counters, int numThreads)
{
    if (mode >= DBG_NUM_MODES) return -1;
    return renderPassImpl(scene, params, width, height, 0, 1, sum, secondary, counters, numThreads, 2 + (int)mode);
}
static int renderPassImpl(const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height,
                          uint32_t shardRank, uint32_t shardWorld, float* sum, float* secondary, uint64_t* counters, int numThreads, int integrator)
{
    if (!scene || !params || !sum || !counters || width == 0 || height == 0) return -1;
    if (numThreads <= 1)
    {
        Counters c; memset(&c, 0, sizeof(c));
        renderRows(scene, params, width, height, shardRank, shardWorld, 0, height, sum, secondary, &c, integrator);
        for (int i = 0; i < 16; ++i) counters[i] += c.c[i];
        return 0;
    }
    std::vector<std::thread> threads;
    std::vector<Counters> cs((size_t)numThreads);
    std::atomic<uint32_t> nextRow(0);
    const uint32_t chunk = 4;
    for (int t = 0; t < numThreads; ++t)
    {
        memset(&cs[(size_t)t], 0, sizeof(Counters));
        threads.emplace_back([&, t]() {
            for (;;)
            {
                const uint32_t r = nextRow.fetch_add(chunk);
                if (r >= height) break;
                const uint32_t r1 = r + chunk < height ? r + chunk : height;
                renderRows(scene, params, width, height, shardRank, shardWorld, r, r1, sum, secondary, &cs[(size_t)t], integrator);
            }
        });
    }
    for (auto& th : threads) th.join();
    for (int t = 0; t < numThreads; ++t) for (int i = 0; i < 16; ++i) counters[i] += cs[(size_t)t].c[i];
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Known-answer-test dispatcher: evaluates one restated function on n records.
// in: n * inStride floats (uint32 payloads are bit-cast), out: n * outStride floats.
// The record layouts are the ones oracle/ref_harness/kat_gen.cpp writes (tests/golden/README.md).
// ---------------------------------------------------------------------------------------------------
enum
{
    KAT_SIN_LANE = 1, KAT_SINCOS = 2, KAT_FASTLOG = 3, KAT_FASTACOS = 4, KAT_FASTATAN2 = 5,
    KAT_FLOAT_NORMAL2 = 6, KAT_HEMISPHERE_COS = 7, KAT_SPHERE = 8, KAT_CIRCLE = 9, KAT_ORTHO_BASIS = 10,
    KAT_FRESNEL_DIELECTRIC = 11, KAT_FRESNEL_METAL = 12, KAT_REFRACT3 = 13, KAT_REFLECT3 = 14,
    KAT_BOX_RAY = 20, KAT_BOX_RAY_TWOSIDED = 21, KAT_TRIANGLE_RAY = 22, KAT_MAKE_RAY = 23, KAT_TRANSFORM_RAY = 24,
    KAT_FAST_INVERSE = 25, KAT_TRANSFORM_SCALED = 26, KAT_FRAME_COMPOSE = 27,
    KAT_SHAPE_INTERSECT = 30, KAT_SHAPE_SAMPLE = 31, KAT_SHAPE_PDF = 32, KAT_SHAPE_EVAL = 33,
    KAT_LIGHT_ILLUMINATE = 40, KAT_LIGHT_RADIANCE = 41,
    KAT_BSDF_SAMPLE = 50, KAT_BSDF_EVALUATE = 51,
    KAT_CAMERA_RAY = 60,
    KAT_LIGHT_EMIT = 42, KAT_LIGHT_ILLUMINATE_BIDIR = 43, KAT_LIGHT_RADIANCE_BIDIR = 44, KAT_BSDF_PDFS = 52,
    KAT_CAMERA_FILM = 61, KAT_FILM_SPLAT = 62, KAT_PACKED_PHOTON = 63, KAT_HSV_TO_RGB = 64,
    KAT_SAMPLER = 70,
};

static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static void putV4(float* o, V4 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }

// the BSDF fixtures carry the scalar part of RtMaterial (through `bsdf`, 52 bytes) in 16 float slots
static RtMaterial katMaterial(const float* in)
{
    RtMaterial m; memset(&m, 0, sizeof(m));
    memcpy(&m, in, 52);
    m.baseColorTexture = m.emissionTexture = m.roughnessTexture = m.metalnessTexture = m.normalMapTexture = RT_NO_TEXTURE;
    return m;
}

int rto_kat(int func, const float* in, int inStride, float* out, int outStride, int n)
{
    for (int r = 0; r < n; ++r)
    {
        const float* i = in + (size_t)r * inStride;
        float* o = out + (size_t)r * outStride;
        switch (func)
        {
        case KAT_SIN_LANE: o[0] = sinLane(i[0]); break;
        case KAT_SINCOS: putV4(o, sinCos(i[0])); break;
        case KAT_FASTLOG: o[0] = fastLog(i[0]); break;
        case KAT_FASTACOS: o[0] = fastACos(i[0]); break;
        case KAT_FASTATAN2: o[0] = fastATan2(i[0], i[1]); break;
        case KAT_FLOAT_NORMAL2: putV4(o, getFloatNormal2(i[0], i[1])); break;
        case KAT_HEMISPHERE_COS: putV4(o, getHemisphereCos(i[0], i[1])); break;
        case KAT_SPHERE: putV4(o, getSphere(i[0], i[1])); break;
        case KAT_CIRCLE: putV4(o, getCircle(i[0], i[1])); break;
        case KAT_ORTHO_BASIS: { V4 u, v; buildOrthonormalBasis(load4(i), u, v); putV4(o, u); putV4(o + 4, v); break; }
        case KAT_FRESNEL_DIELECTRIC: o[0] = fresnelDielectric(i[0], i[1]); break;
        case KAT_FRESNEL_METAL: o[0] = fresnelMetal(i[0], i[1], i[2]); break;
        case KAT_REFRACT3: putV4(o, refract3(load4(i), load4(i + 4), i[8])); break;
        case KAT_REFLECT3: putV4(o, reflect3(load4(i), load4(i + 4))); break;
        case KAT_MAKE_RAY:
        {
            const Ray ray = makeRay(load4(i), load4(i + 4));
            putV4(o, ray.dir); putV4(o + 4, ray.invDir); putV4(o + 8, ray.originDivDir); break;
        }
        case KAT_TRANSFORM_RAY:   // in: matrix[16], origin[4], dir[4] (normalized world ray)  out: origin, dir, invDir, originDivDir
        {
            const M4 m = loadM4(i);
            Ray w; w.origin = load4(i + 16); w.dir = load4(i + 20); w.invDir = zero4(); w.originDivDir = zero4();
            const Ray l = transformRayUnsafe(m, w);
            putV4(o, l.origin); putV4(o + 4, l.dir); putV4(o + 8, l.invDir); putV4(o + 12, l.originDivDir); break;
        }
        case KAT_FAST_INVERSE: { const M4 m = fastInverseNoScale(loadM4(i)); for (int k = 0; k < 4; ++k) putV4(o + 4 * k, m.r[k]); break; }
        case KAT_TRANSFORM_SCALED:   // in: matrix[16] (rotation x scale + translation), v[4]  out: TransformPoint, TransformVector, FastInverseNoScale().TransformPoint
        {
            const M4 m = loadM4(i); const V4 v = load4(i + 16);
            putV4(o, transformPoint(m, v)); putV4(o + 4, transformVector(m, v)); putV4(o + 8, transformPoint(fastInverseNoScale(m), v)); break;
        }
        case KAT_FRAME_COMPOSE:      // Scene::EvaluateIntersection's frame, Scene.cpp:311-348 (layout: kat_gen.cpp)
        {
            const M4 transform = loadM4(i);
            Ray ray; ray.origin = load4(i + 16); ray.dir = load4(i + 20); ray.invDir = zero4(); ray.originDivDir = zero4();
            const V4 worldPosition = rayAt(ray, i[24]);
            M4 frame;
            composeShadingFrame(transform, worldPosition, load4(i + 28), load4(i + 32), i[25] != 0.0f, load4(i + 36), frame);
            putV4(o, transformPoint(fastInverseNoScale(transform), worldPosition));
            for (int k = 0; k < 4; ++k) putV4(o + 4 + 4 * k, frame.r[k]);
            break;
        }
        case KAT_BOX_RAY:         // in: origin[4], direction[4] (unnormalized), bmin[3], bmax[3]
        {
            const Ray ray = makeRay(load4(i), load4(i + 4));
            float d = 0.0f; const bool h = intersectBoxRay(ray, load3(i + 8), load3(i + 11), d);
            o[0] = bitsf(h ? 1u : 0u); o[1] = d; break;
        }
        case KAT_BOX_RAY_TWOSIDED:
        {
            const Ray ray = makeRay(load4(i), load4(i + 4));
            float a = 0.0f, b = 0.0f; const bool h = intersectBoxRayTwoSided(ray, load3(i + 8), load3(i + 11), a, b);
            o[0] = bitsf(h ? 1u : 0u); o[1] = a; o[2] = b; break;
        }
        case KAT_TRIANGLE_RAY:    // in: origin[4], direction[4], v0[3], e1[3], e2[3]
        {
            const Ray ray = makeRay(load4(i), load4(i + 4));
            float u = 0, v = 0, t = 0; const bool h = intersectTriangleRay(ray, load3(i + 8), load3(i + 11), load3(i + 14), u, v, t);
            o[0] = bitsf(h ? 1u : 0u); o[1] = u; o[2] = v; o[3] = t; break;
        }
        case KAT_SHAPE_INTERSECT: // in: kind(bits), param[4], origin[4], direction[4]
        {
            const Ray ray = makeRay(load4(i + 5), load4(i + 9));
            ShapeHit sh; sh.nearDist = 0; sh.farDist = 0;
            const bool h = shapeIntersect(fbits(i[0]), i + 1, ray, sh);
            o[0] = bitsf(h ? 1u : 0u); o[1] = h ? sh.nearDist : 0.0f; o[2] = h ? sh.farDist : 0.0f; o[3] = bitsf(sh.subObjectId); break;
        }
        case KAT_SHAPE_SAMPLE:    // in: kind, param[4], ref[4], u[3]
        {
            ShapeSample s; s.direction = zero4(); s.distance = s.pdf = s.cosAtSurface = -1.0f;
            const bool h = shapeSampleFrom(fbits(i[0]), i + 1, load4(i + 5), i + 9, s);
            o[0] = bitsf(h ? 1u : 0u);
            if (h) { putV4(o + 1, s.direction); o[5] = s.distance; o[6] = s.pdf; o[7] = s.cosAtSurface; }
            else { for (int k = 1; k < 8; ++k) o[k] = 0.0f; }
            break;
        }
        case KAT_SHAPE_PDF: o[0] = shapePdf(fbits(i[0]), i + 1, load4(i + 5), load4(i + 9)); break;
        case KAT_SHAPE_EVAL:      // in: kind, param[4], param2[4], localPos[4]   out: frame rows 0..2, texCoord
        {
            Intersection is; for (int k = 0; k < 4; ++k) is.frame.r[k] = zero4();
            is.frame.r[3] = load4(i + 9); is.texCoord = zero4(); is.material = 0;
            shapeEvaluateIntersection(fbits(i[0]), i + 1, i + 5, is);
            putV4(o, is.frame.r[0]); putV4(o + 4, is.frame.r[1]); putV4(o + 8, is.frame.r[2]); putV4(o + 12, is.texCoord); break;
        }
        case KAT_LIGHT_ILLUMINATE: // in: RtLight as floats (sizeof/4), frame[16], u[3]
        {
            const int LW = (int)(sizeof(RtLight) / 4);
            RtLight L; memcpy(&L, i, sizeof(RtLight));
            Intersection is; is.frame = loadM4(i + LW); is.texCoord = zero4(); is.material = 0;
            IlluminateResult ir;
            const V4 rad = lightIlluminate(nullptr, L, is, i + LW + 16, ir);
            putV4(o, rad); putV4(o + 4, ir.directionToLight); o[8] = ir.distance; o[9] = ir.directPdfW; o[10] = ir.cosAtLight; break;
        }
        case KAT_LIGHT_RADIANCE:  // in: RtLight, ray origin[4], dir[4] (light space), hitPoint[4], cosAtLight
        {
            const int LW = (int)(sizeof(RtLight) / 4);
            RtLight L; memcpy(&L, i, sizeof(RtLight));
            Ray ray; ray.origin = load4(i + LW); ray.dir = load4(i + LW + 4); ray.invDir = zero4(); ray.originDivDir = zero4();
            float pdf = 0.0f;
            const V4 rad = lightGetRadiance(nullptr, L, ray, load4(i + LW + 8), i[LW + 12], pdf);
            putV4(o, rad); o[4] = pdf; break;
        }
        case KAT_BSDF_SAMPLE:     // in: the first 64 bytes of RtMaterial (16 floats; no textures), outgoingDir[4] (local), u[3]
        {
            RtMaterial m = katMaterial(i);
            ShadingData sd; sd.intersection.texCoord = zero4(); materialEvaluateShadingData(nullptr, m, sd);
            BsdfSample s;
            const bool ok = bsdfSampleImpl(m.bsdf, m, sd.mp, i + 20, load4(i + 16), s);
            o[0] = bitsf(ok ? 1u : 0u);
            if (ok) { putV4(o + 1, s.color); putV4(o + 5, s.incomingDir); o[9] = s.pdf; o[10] = bitsf(s.event); }
            else { for (int k = 1; k < 11; ++k) o[k] = 0.0f; }
            break;
        }
        case KAT_BSDF_EVALUATE:   // in: RtMaterial, outgoingDir[4], incomingDir[4] (local)
        {
            RtMaterial m = katMaterial(i);
            ShadingData sd; sd.intersection.texCoord = zero4(); materialEvaluateShadingData(nullptr, m, sd);
            float pdf = 0.0f;
            const V4 c = bsdfEvaluate(m.bsdf, m, sd.mp, load4(i + 16), load4(i + 20), pdf);
            putV4(o, c); o[4] = almostZero4(c) ? 0.0f : pdf; break;
        }
        case KAT_CAMERA_RAY:      // in: RtCamera (sizeof/4 floats), coords[2], dof samples from seed {u0,u1} bits, Random::mSeed[2] (barrel distortion)
        {
            const int CW = (int)(sizeof(RtCamera) / 4);
            RtCamera cam; memcpy(&cam, i, sizeof(RtCamera));
            uint32_t seeds[2] = { fbits(i[CW + 2]), fbits(i[CW + 3]) };
            Sampler s; s.seed = seeds; s.numDims = 2; s.blueNoiseLayers = 0; s.blueNoise = nullptr;
            s.bx = s.by = 0; s.salt = 0; s.generated = 0;
            memcpy(&s.fallback.s[0], i + CW + 4, 8); memcpy(&s.fallback.s[1], i + CW + 6, 8);   // Random::mSeed of ctx.randomGenerator
            const Ray ray = cameraGenerateRay(cam, V4(i[CW], i[CW + 1], 0.0f, 0.0f), s);
            putV4(o, ray.origin); putV4(o + 4, ray.dir); putV4(o + 8, ray.invDir); putV4(o + 12, ray.originDivDir); break;
        }
        case KAT_LIGHT_EMIT:      // in: RtLight, positionSample[3], directionSample[2]
        {
            const int LW = (int)(sizeof(RtLight) / 4);
            RtLight L; memcpy(&L, i, sizeof(RtLight));
            EmitResult er; er.position = zero4(); er.direction = zero4(); er.directPdfA = er.emissionPdfW = er.cosAtLight = 0.0f;
            const V4 c = lightEmit(nullptr, L, i + LW, i + LW + 3, er);
            putV4(o, c); putV4(o + 4, er.position); putV4(o + 8, er.direction); o[12] = er.directPdfA; o[13] = er.emissionPdfW; o[14] = er.cosAtLight; break;
        }
        case KAT_LIGHT_ILLUMINATE_BIDIR: // in: RtLight, frame[16], u[3]; rendererSupportsSolidAngleSampling = false
        {
            const int LW = (int)(sizeof(RtLight) / 4);
            RtLight L; memcpy(&L, i, sizeof(RtLight));
            Intersection is; is.frame = loadM4(i + LW); is.texCoord = zero4(); is.material = 0;
            IlluminateResult ir;
            const V4 rad = lightIlluminate(nullptr, L, is, i + LW + 16, ir, false);
            putV4(o, rad); putV4(o + 4, ir.directionToLight); o[8] = ir.distance; o[9] = ir.directPdfW; o[10] = ir.emissionPdfW; o[11] = ir.cosAtLight; break;
        }
        case KAT_LIGHT_RADIANCE_BIDIR:
        {
            const int LW = (int)(sizeof(RtLight) / 4);
            RtLight L; memcpy(&L, i, sizeof(RtLight));
            Ray ray; ray.origin = load4(i + LW); ray.dir = load4(i + LW + 4); ray.invDir = zero4(); ray.originDivDir = zero4();
            float pdfA = 0.0f, pdfW = 0.0f;
            const V4 rad = lightGetRadiance(nullptr, L, ray, load4(i + LW + 8), i[LW + 12], pdfA, &pdfW, false);
            putV4(o, rad); o[4] = almostZero4(rad) ? 0.0f : pdfA; o[5] = almostZero4(rad) ? 0.0f : pdfW; break;
        }
        case KAT_BSDF_PDFS:       // in: RtMaterial, outgoingDir[4], incomingDir[4]   out: colour, pdf, reverse pdf, Pdf(Forward), Pdf(Reverse)
        {
            RtMaterial m = katMaterial(i);
            ShadingData sd; sd.intersection.texCoord = zero4(); materialEvaluateShadingData(nullptr, m, sd);
            float pdf = 0.0f, rev = 0.0f;
            const V4 c = bsdfEvaluate(m.bsdf, m, sd.mp, load4(i + 16), load4(i + 20), pdf, &rev);
            putV4(o, c); o[4] = almostZero4(c) ? 0.0f : pdf; o[5] = almostZero4(c) ? 0.0f : rev;
            o[6] = bsdfPdf(m.bsdf, m, sd.mp, load4(i + 16), load4(i + 20), false);
            o[7] = bsdfPdf(m.bsdf, m, sd.mp, load4(i + 16), load4(i + 20), true); break;
        }
        case KAT_CAMERA_FILM:     // in: RtCamera, world position[4], direction[4]   out: visible, film coords[4], PdfW
        {
            const int CW = (int)(sizeof(RtCamera) / 4);
            RtCamera cam; memcpy(&cam, i, sizeof(RtCamera));
            V4 film = zero4();
            const bool ok = cameraWorldToFilm(cam, load4(i + CW), film);
            o[0] = bitsf(ok ? 1u : 0u); o[1] = ok ? film.x : 0.0f; o[2] = ok ? film.y : 0.0f; o[3] = ok ? film.z : 0.0f; o[4] = ok ? film.w : 0.0f;
            o[5] = cameraDirectionPdfW(cam, load4(i + CW + 4)); break;
        }
        case KAT_FILM_SPLAT:      // in: pos[2], width, height, mSeedSimd4[0..1]   out: x, y (0xFFFFFFFF = outside), generator state after
        {
            RandomSimd rng; memcpy(rng.seed0, i + 4, 16); memcpy(rng.seed1, i + 8, 16);
            uint32_t x = 0xFFFFFFFFu, y = 0xFFFFFFFFu;
            if (!filmSplatPixel(V4(i[0], i[1], 0.0f, 0.0f), fbits(i[2]), fbits(i[3]), rng.getVector4(), x, y)) { x = y = 0xFFFFFFFFu; }
            o[0] = bitsf(x); o[1] = bitsf(y); memcpy(o + 2, rng.seed0, 16); memcpy(o + 6, rng.seed1, 16); break;
        }
        case KAT_HSV_TO_RGB: putV4(o, debugTriangleIdColor(fbits(i[0]), fbits(i[1]))); break;
        case KAT_PACKED_PHOTON:   // in: direction[4], colour[4]   out: packed direction, packed colour (2 words), unpacked direction[4], colour[4]
        {
            const uint32_t pd = packUnitVector(load4(i));
            const PackedColor pc = packColorHdr(load4(i + 4));
            o[0] = bitsf(pd); memcpy(o + 1, &pc, 8);
            putV4(o + 3, unpackUnitVector(pd)); putV4(o + 7, unpackColorHdr(pc)); break;
        }
        default: return -1;
        }
    }
    return 0;
}

// Sampler KAT: ints for dims [0, count) at pixel (x, y).
int rto_kat_sampler(const uint32_t* seed, uint32_t numDims, const uint16_t* blueNoise, uint32_t useBlueNoise,
                    uint32_t x, uint32_t y, uint32_t count, uint32_t* outInts, float* outFloats)
{
    Sampler s; s.seed = seed; s.numDims = numDims; s.blueNoise = blueNoise;
    s.blueNoiseLayers = (blueNoise && useBlueNoise) ? 4u : 0u;
    const uint64_t key[2] = { 0, 0 };
    s.resetPixel(x, y, key);
    for (uint32_t i = 0; i < count; ++i)
    {
        Sampler copy = s;
        outInts[i] = s.getInt();
        if (outFloats) outFloats[i] = copy.getFloat();
    }
    return 0;
}

// xoroshiro128+ KAT (Random::GetLong / GetFloat, Core/Math/Random.cpp:33-59)
int rto_kat_xoroshiro(uint64_t s0, uint64_t s1, uint32_t count, uint64_t* outLongs)
{
    Xoroshiro g; g.s[0] = s0; g.s[1] = s1;
    for (uint32_t i = 0; i < count; ++i) outLongs[i] = xoroshiroNext(g);
    return 0;
}

// Per-pixel debug trace: renders ONE pixel and returns its radiance (for path-level parity debugging).  rto_render_pixel_paths also
// records the path's vertices like the reference's PathDebugData hook does (28 floats per vertex, see RenderCtx); *numVertices = how many
// the path has (only the first `capacity` are stored).
static float* gPathDump = nullptr; static uint32_t gPathDumpCapacity = 0; static uint32_t* gPathDumpCount = nullptr;
int rto_render_pixel(const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height,
                     uint32_t x, uint32_t y, float outRGBA[4], uint64_t* counters);
int rto_render_pixel_paths(const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height, uint32_t x, uint32_t y,
                           float outRGBA[4], uint64_t* counters, float* vertices, uint32_t capacity, uint32_t* numVertices)
{
    gPathDump = vertices; gPathDumpCapacity = capacity; gPathDumpCount = numVertices;
    const int r = rto_render_pixel(scene, params, width, height, x, y, outRGBA, counters);
    gPathDump = nullptr; gPathDumpCapacity = 0; gPathDumpCount = nullptr;
    return r;
}
int rto_render_pixel(const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height,
                     uint32_t x, uint32_t y, float outRGBA[4], uint64_t* counters)
{
    Counters c; memset(&c, 0, sizeof(c));
    RenderCtx ctx;
    ctx.pathDump = gPathDump; ctx.pathDumpCapacity = gPathDumpCapacity;
    ctx.scene = scene; ctx.params = params; ctx.counters = &c;
    ctx.lightSamplingWeight = load4(params->lightSamplingWeight);
    ctx.bsdfSamplingWeight = load4(params->bsdfSamplingWeight);
    ctx.sampler.seed = params->seed; ctx.sampler.numDims = params->numDimensions; ctx.sampler.blueNoise = scene->blueNoise;
    ctx.sampler.blueNoiseLayers = (scene->blueNoise && params->useBlueNoise) ? 4u : 0u;
    const V4 invSize(1.0f / (float)(int32_t)width, 1.0f / (float)(int32_t)height, 0.0f, 0.0f);
    const V4 sampleOffset(params->sampleOffset[0], params->sampleOffset[1], 0.0f, 0.0f);
    const uint32_t realY = height - 1u - y;
    const V4 coords = (V4((float)(int32_t)x, (float)(int32_t)realY, 0.0f, 0.0f) + sampleOffset) * invSize;
    ctx.sampler.resetPixel(x, y, params->rngKey);
    const Ray ray = cameraGenerateRay(params->camera, coords, ctx.sampler);
    const V4 color = renderPixel(ctx, ray);
    if (gPathDumpCount) *gPathDumpCount = ctx.pathDumpCount;
    outRGBA[0] = color.x; outRGBA[1] = color.y; outRGBA[2] = color.z; outRGBA[3] = color.w;
    if (counters) for (int i = 0; i < 16; ++i) counters[i] += c.c[i];
    return 0;
}

// Mesh-path KAT (layout of tests/golden/mesh_kat.bin, written by oracle/ref_harness/kat_gen.cpp::genMesh):
// rays: n * 7 floats (origin, direction, tmax); out: n * 19 uint32 words
//   [objectId, subObjectId, distance, u, v, shadowHit, frame0.xyzw, frame2.xyzw, texCoord.xyzw, material]
// Calls the restated MeshShape::Traverse / Traverse_Shadow / EvaluateIntersection on mesh `meshIndex`
// with objectID = 7, exactly like the generator does with the reference's MeshShape.
int rto_kat_mesh(const RtSceneDesc* scene, uint32_t meshIndex, const float* rays, uint32_t n, uint32_t* out)
{
    if (!scene || meshIndex >= scene->numMeshes) return -1;
    const RtMesh& mesh = scene->meshes[meshIndex];
    Counters cnt; memset(&cnt, 0, sizeof(cnt));
    for (uint32_t i = 0; i < n; ++i)
    {
        const float* r = rays + 7 * (size_t)i;
        uint32_t* o = out + 19 * (size_t)i;
        const Ray ray = makeRay(V4(r[0], r[1], r[2], 0.0f), V4(r[3], r[4], r[5], 0.0f));
        Hit hp; hp.objectId = RT_INVALID_OBJECT; hp.subObjectId = 0; hp.distance = r[6]; hp.u = 0.0f; hp.v = 0.0f;
        meshTraverse(scene, mesh, ray, hp, 7, cnt);
        const bool hit = hp.objectId == 7;
        o[0] = hp.objectId; o[1] = hit ? hp.subObjectId : 0u; o[2] = fbits(hp.distance); o[3] = fbits(hit ? hp.u : 0.0f); o[4] = fbits(hit ? hp.v : 0.0f);
        Hit hs; hs.objectId = RT_INVALID_OBJECT; hs.subObjectId = 0; hs.distance = r[6]; hs.u = hs.v = 0.0f;
        o[5] = meshTraverseShadow(scene, mesh, ray, hs, cnt) ? 1u : 0u;
        for (int k = 6; k < 18; ++k) o[k] = 0;
        o[18] = 0xFFFFFFFFu;
        if (hit)
        {
            Intersection is; for (int k = 0; k < 4; ++k) is.frame.r[k] = zero4();
            is.texCoord = zero4(); is.material = RT_NO_MATERIAL;
            meshEvaluateIntersection(scene, mesh, hp, is);
            const V4 f0 = is.frame.r[0], f2 = is.frame.r[2], tc = is.texCoord;
            o[6] = fbits(f0.x); o[7] = fbits(f0.y); o[8] = fbits(f0.z); o[9] = fbits(f0.w);
            o[10] = fbits(f2.x); o[11] = fbits(f2.y); o[12] = fbits(f2.z); o[13] = fbits(f2.w);
            o[14] = fbits(tc.x); o[15] = fbits(tc.y); o[16] = fbits(tc.z); o[17] = fbits(tc.w);
            o[18] = is.material;
        }
    }
    return 0;
}

// ---- textures on the shading path (tests/golden/texture_kat.bin) -------------------------------------------------
// ITexture::Evaluate of textures[index] at (u, v)
void rto_texture_evaluate(const RtTexture* textures, const uint8_t* texels, uint32_t index, float u, float v, float out[4])
{
    RtSceneDesc d; memset(&d, 0, sizeof(d)); d.textures = textures; d.texelData = texels;
    const V4 c = textureEvaluate(&d, index, V4(u, v, 0.0f, 0.0f));
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}
// Material::EvaluateShadingData + GetNormalVector: out = baseColor[4], emission[4], roughness, metalness, normal[4]
void rto_material_shading(const RtTexture* textures, const uint8_t* texels, const RtMaterial* mat, float u, float v, float out[14])
{
    RtSceneDesc d; memset(&d, 0, sizeof(d)); d.textures = textures; d.texelData = texels;
    ShadingData sd; sd.intersection.texCoord = V4(u, v, 0.0f, 0.0f);
    materialEvaluateShadingData(&d, *mat, sd);
    putV4(out, sd.mp.baseColor); putV4(out + 4, sd.mp.emission); out[8] = sd.mp.roughness; out[9] = sd.mp.metalness;
    V4 n(0.0f, 0.0f, 1.0f, 0.0f);
    if (mat->normalMapTexture != RT_NO_TEXTURE) n = materialGetNormalVector(&d, *mat, sd.intersection.texCoord);
    putV4(out + 10, n);
}
// BackgroundLight::GetRadiance with an environment map
void rto_background_radiance(const RtTexture* textures, const uint8_t* texels, const RtLight* light, const float dir[4], float out[4])
{
    RtSceneDesc d; memset(&d, 0, sizeof(d)); d.textures = textures; d.texelData = texels;
    Ray ray; ray.origin = zero4(); ray.dir = load4(dir); ray.invDir = zero4(); ray.originDivDir = zero4();
    float pdf = 0.0f;
    putV4(out, lightGetRadiance(&d, *light, ray, zero4(), 1.0f, pdf));
}

// Viewport::PostProcessTile over a whole float3 sum buffer -> 0x00RRGGBB front buffer
void rto_postprocess(const float* sumRGB, uint32_t width, uint32_t height, const RtPostprocessParams* params, uint32_t* out)
{
    const float exposureScale = powf(2.0f, params->exposure);
    const float colorScale[3] = { params->colorFilter[0] * exposureScale, params->colorFilter[1] * exposureScale, params->colorFilter[2] * exposureScale };
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x)
        {
            const float* px = sumRGB + 3 * ((size_t)y * width + x);
            out[(size_t)y * width + x] = postProcessPixel(px[0], px[1], px[2], x, y, *params, colorScale);
        }
}

// ---- bloom -------------------------------------------------------------------------------------------------------
// BoxBlur_Internal, Core/Utils/Bitmap.cpp:880-914 (one colour channel; the reference runs the four lanes of a Vector4 in lockstep)
static void boxBlurInternal(float* targetLine, const float* srcLine, const uint32_t radius, const uint32_t width)
{
    const float factor = 1.0f / (float)(2 * radius + 1);
    const float* srcLineBegin = srcLine; const float* srcLineEnd = srcLine;
    const float firstValue = srcLine[0];
    const float lastValue = srcLine[width - 1];
    float val = firstValue * (float)(radius + 1);
    for (uint32_t j = 0; j < radius; j++) val = val + *(srcLineBegin++);
    for (uint32_t j = 0; j <= radius; j++) { val = val + (*(srcLineBegin++) - firstValue); *(targetLine++) = val * factor; }
    for (uint32_t j = radius + 1; j < width - radius; j++) { val = val + (*(srcLineBegin++) - *(srcLineEnd++)); *(targetLine++) = val * factor; }
    for (uint32_t j = width - radius; j < width; j++) { val = val + (lastValue - *(srcLineEnd++)); *(targetLine++) = val * factor; }
}
// Bitmap::GaussianBlur, :917-1020, on a tight float3 image.  Lines are processed per channel; the line buffers are as long as the
// reference's (4096) so that a window wider than the line reads the same kind of stale entries -- callers keep to sizes where it does not.
int rto_gaussian_blur(float* rgb, uint32_t width, uint32_t height, float sigma, uint32_t n)
{
    const uint32_t MaxLineSize = 4096;
    if (width > MaxLineSize || height > MaxLineSize) return -1;
    float wIdeal = sqrtf((12.0f * sigma * sigma / n) + 1.0f);
    uint32_t wl = (uint32_t)floorf(wIdeal);
    if (wl % 2 == 0) wl--;
    const uint32_t wu = wl + 2;
    const float mIdeal = (12.0f * sigma * sigma - n * wl * wl - 4.0f * n * wl - 3.0f * n) / (-4.0f * wl - 4.0f);
    const float m = roundf(mIdeal);
    std::vector<float> tempA(MaxLineSize, 0.0f), tempB(MaxLineSize, 0.0f);
    for (int channel = 0; channel < 3; ++channel)
    {
        for (uint32_t y = 0; y < height; ++y)   // horizontal blur, :943-965
        {
            float* rowPtr = rgb + 3 * (size_t)y * width + channel;
            for (uint32_t x = 0; x < width; ++x) tempB[x] = rowPtr[3 * (size_t)x];
            float* sourceLinePtr = tempB.data(); float* targetLinePtr = tempA.data();
            for (uint32_t i = 0; i < n; ++i)
            {
                const uint32_t radius = i < m ? wl : wu;
                boxBlurInternal(targetLinePtr, sourceLinePtr, radius, width);
                std::swap(sourceLinePtr, targetLinePtr);
            }
            for (uint32_t x = 0; x < width; ++x) rowPtr[3 * (size_t)x] = targetLinePtr[x];   // as written in the reference: the buffer the LAST blur read from
        }
    }
    for (int channel = 0; channel < 3; ++channel)
    {
        for (uint32_t x = 0; x < width; ++x)    // vertical blur, :968-1011 (four columns at a time there; columns are independent)
        {
            float* colPtr = rgb + 3 * (size_t)x + channel;
            for (uint32_t y = 0; y < height; ++y) tempA[y] = colPtr[3 * (size_t)y * width];
            float* sourceLinePtr = tempA.data(); float* targetLinePtr = tempB.data();
            for (uint32_t j = 0; j < n; ++j)
            {
                const uint32_t radius = j < m ? wl : wu;
                boxBlurInternal(targetLinePtr, sourceLinePtr, radius, height);
                std::swap(sourceLinePtr, targetLinePtr);
            }
            for (uint32_t y = 0; y < height; ++y) colPtr[3 * (size_t)y * width] = tempA[y];
        }
    }
    return 0;
}

// Viewport::PerformPostProcess + PostProcessTile with bloom (Viewport.cpp:432-452, 512-524)
int rto_postprocess_bloom(const float* sumRGB, uint32_t width, uint32_t height, const RtPostprocessParams* params, uint32_t* out)
{
    const size_t count = (size_t)width * height * 3;
    std::vector<std::vector<float>> blurred(5);
    float blurSigma = 2.0f;
    for (int i = 0; i < 5; ++i)
    {
        blurred[i].assign(i == 0 ? sumRGB : blurred[i - 1].data(), (i == 0 ? sumRGB : blurred[i - 1].data()) + count);
        if (rto_gaussian_blur(blurred[i].data(), width, height, blurSigma, 8) != 0) return -1;
        blurSigma *= 2.5f;
    }
    const float bloomWeights[] = { 0.35f, 0.25f, 0.15f, 0.15f, 0.1f };
    const float exposureScale = powf(2.0f, params->exposure);
    const float colorScale[3] = { params->colorFilter[0] * exposureScale, params->colorFilter[1] * exposureScale, params->colorFilter[2] * exposureScale };
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x)
        {
            const size_t i = (size_t)y * width + x;
            float rgb[3];
            for (int k = 0; k < 3; ++k)
            {
                const float v = sumRGB[3 * i + k] * (1.0f - params->bloomFactor);
                float bloomColor = 0.0f;
                for (int l = 0; l < 5; ++l) bloomColor = fmaf(blurred[l][3 * i + k], bloomWeights[l], bloomColor);
                rgb[k] = fmaf(bloomColor, params->bloomFactor, v);
            }
            out[i] = postProcessPixel(rgb[0], rgb[1], rgb[2], x, y, *params, colorScale);
        }
    return 0;
}

// Viewport::ComputeBlockError, Core/Rendering/Viewport.cpp:552-581
float rto_block_error(const float* sumRGB, const float* secondaryRGB, uint32_t width, uint32_t height, uint32_t numPasses,
                      uint32_t minX, uint32_t maxX, uint32_t minY, uint32_t maxY)
{
    const float imageScalingFactor = 1.0f / (float)numPasses;
    float totalError = 0.0f;
    for (uint32_t y = minY; y < maxY; ++y)
    {
        float rowError = 0.0f;
        for (uint32_t x = minX; x < maxX; ++x)
        {
            const V4 a = imageScalingFactor * load3(sumRGB + 3 * ((size_t)y * width + x));
            const V4 b = (2.0f * imageScalingFactor) * load3(secondaryRGB + 3 * ((size_t)y * width + x));
            const V4 diff = abs4(a - b);
            const float error = (diff.x + 2.0f * diff.y + diff.z) / sqrtf(RTO_EPSILON + a.x + 2.0f * a.y + a.z);
            rowError += error;
        }
        totalError += rowError;
    }
    const uint32_t totalArea = width * height, blockArea = (maxX - minX) * (maxY - minY);
    return totalError * sqrtf((float)blockArea / (float)totalArea) / (float)blockArea;
}

// ---- bidirectional integrator (VertexConnectionAndMerging) -- see rto_vcm.h for the parity status ------------------------
// settings: 8 words { maxPathLength, useVertexConnection, useVertexMerging, initialMergingRadius, minMergingRadius,
// mergingRadiusMultiplier, 0, 0 } followed by the five float4 weights (bsdf, light, vertexConnecting, cameraConnecting, merging)
void* rto_vcm_create(const uint32_t* settingsWords)
{
    VcmRenderer* r = new VcmRenderer();
    if (settingsWords)
    {
        r->s.maxPathLength = settingsWords[0]; r->s.useVertexConnection = settingsWords[1]; r->s.useVertexMerging = settingsWords[2];
        r->s.initialMergingRadius = bitsf(settingsWords[3]); r->s.minMergingRadius = bitsf(settingsWords[4]); r->s.mergingRadiusMultiplier = bitsf(settingsWords[5]);
        memcpy(r->s.bsdfSamplingWeight, settingsWords + 8, 16); memcpy(r->s.lightSamplingWeight, settingsWords + 12, 16);
        memcpy(r->s.vertexConnectingWeight, settingsWords + 16, 16); memcpy(r->s.cameraConnectingWeight, settingsWords + 20, 16);
        memcpy(r->s.vertexMergingWeight, settingsWords + 24, 16);
    }
    return r;
}
// HashGrid::Build + HashGrid::Process over plain points (the subject of the reference's own Tests/HashGridTest.cpp): for every query the
// particle indices in visiting order, appended to outIndices (capacity `capacity`); outOffsets[q]..outOffsets[q+1] delimit query q.
// Returns the total number of indices (which may exceed the capacity: then only the first `capacity` were written).
uint64_t rto_hash_grid_query(const float* points, uint32_t numPoints, float radius, const float* queries, uint32_t numQueries,
                             uint64_t* outOffsets, uint32_t* outIndices, uint64_t capacity)
{
    std::vector<Photon> particles(numPoints);
    for (uint32_t i = 0; i < numPoints; ++i) { memset(&particles[i], 0, sizeof(Photon)); particles[i].position[0] = points[3 * i]; particles[i].position[1] = points[3 * i + 1]; particles[i].position[2] = points[3 * i + 2]; }
    HashGrid grid; grid.build(particles, radius);
    struct Query { uint32_t* out; uint64_t capacity, total; void operator()(uint32_t index) { if (total < capacity) out[total] = index; total++; } } query = { outIndices, capacity, 0 };
    for (uint32_t q = 0; q < numQueries; ++q)
    {
        outOffsets[q] = query.total;
        grid.process(V4(queries[3 * q], queries[3 * q + 1], queries[3 * q + 2], 0.0f), particles, query);
    }
    outOffsets[numQueries] = query.total;
    return query.total;
}

void rto_vcm_destroy(void* h) { delete static_cast<VcmRenderer*>(h); }
uint32_t rto_vcm_num_photons(void* h) { return (uint32_t)static_cast<VcmRenderer*>(h)->recorded.size(); }

// One pass over the whole film, pixels in row-major order.  Camera-path radiance goes to sum (+ secondarySum when non-null);
// light-path splats go to lightSum when it is non-null (so the two estimators can be compared separately), else to sum.
static int vcmRenderPassImpl(void* h, const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height, uint32_t passNumber,
                             float* sum, float* secondarySum, float* lightSum, uint64_t* counters, uint32_t shardRank, uint32_t shardWorld);
int rto_vcm_render_pass(void* h, const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height, uint32_t passNumber,
                        float* sum, float* secondarySum, float* lightSum, uint64_t* counters)
{
    return vcmRenderPassImpl(h, scene, params, width, height, passNumber, sum, secondarySum, lightSum, counters, 0, 1);
}
// Only the pixels of the 64x64 tiles with tile % shardWorld == shardRank (sampling a large frame).  A pixel's camera-path radiance depends
// on other pixels only through merging, so with merging off (or in pass 0) the owned pixels get exactly their full-frame values; the
// light image and the photon set are those of the owned pixels' light paths only.
int rto_vcm_render_pass_tiles(void* h, const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height, uint32_t passNumber,
                              float* sum, float* secondarySum, float* lightSum, uint64_t* counters, uint32_t shardRank, uint32_t shardWorld)
{
    return vcmRenderPassImpl(h, scene, params, width, height, passNumber, sum, secondarySum, lightSum, counters, shardRank, shardWorld);
}
static int vcmRenderPassImpl(void* h, const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height, uint32_t passNumber,
                             float* sum, float* secondarySum, float* lightSum, uint64_t* counters, uint32_t shardRank, uint32_t shardWorld)
{
    VcmRenderer* r = static_cast<VcmRenderer*>(h);
    r->preRender(passNumber, width, height);
    Counters c; memset(&c, 0, sizeof(c));
    VcmCtx* ctx = new VcmCtx();
    ctx->scene = scene; ctx->params = params; ctx->r = r; ctx->counters = &c;
    ctx->width = width; ctx->height = height;
    ctx->sum = lightSum ? lightSum : sum; ctx->secondarySum = lightSum ? nullptr : secondarySum;
    ctx->sampler.seed = params->seed; ctx->sampler.numDims = params->numDimensions; ctx->sampler.blueNoise = scene->blueNoise;
    ctx->sampler.blueNoiseLayers = (scene->blueNoise && params->useBlueNoise) ? 4u : 0u;
    const V4 invSize(1.0f / (float)(int32_t)width, 1.0f / (float)(int32_t)height, 0.0f, 0.0f);
    const V4 sampleOffset(params->sampleOffset[0], params->sampleOffset[1], 0.0f, 0.0f);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x)
        {
            if (shardWorld > 1 && ((y / 64u) * ((width + 63u) / 64u) + (x / 64u)) % shardWorld != shardRank) continue;
            const uint32_t realY = height - 1u - y;
            const V4 coords = (V4((float)(int32_t)x, (float)(int32_t)realY, 0.0f, 0.0f) + sampleOffset) * invSize;
            ctx->sampler.resetPixel(x, y, params->rngKey);
            ctx->simd.resetPixel(x, y, params->rngKey);
            const Ray ray = cameraGenerateRay(params->camera, coords, ctx->sampler);
            c.c[C_PRIMARY]++;
            const V4 color = vcmRenderPixel(*ctx, ray, passNumber);
            float* p = sum + 3 * ((size_t)y * width + x);
            p[0] = p[0] + color.x; p[1] = p[1] + color.y; p[2] = p[2] + color.z;
            if (secondarySum) { float* q = secondarySum + 3 * ((size_t)y * width + x); q[0] = q[0] + color.x; q[1] = q[1] + color.y; q[2] = q[2] + color.z; }
        }
    delete ctx;
    if (counters) for (int i = 0; i < 16; ++i) counters[i] += c.c[i];
    return 0;
}

// the renderer "Light Tracer" (Core/Rendering/LightTracer.cpp): one pass, pixels in row-major order; all light goes to `sum`
// (+ secondarySum) through film splats
int rto_light_tracer_render_pass(const RtSceneDesc* scene, const RtPassParams* params, uint32_t width, uint32_t height, float* sum, float* secondarySum, uint64_t* counters)
{
    Counters c; memset(&c, 0, sizeof(c));
    VcmRenderer dummy;
    VcmCtx* ctx = new VcmCtx();
    ctx->scene = scene; ctx->params = params; ctx->r = &dummy; ctx->counters = &c;
    ctx->width = width; ctx->height = height; ctx->sum = sum; ctx->secondarySum = secondarySum;
    ctx->sampler.seed = params->seed; ctx->sampler.numDims = params->numDimensions; ctx->sampler.blueNoise = scene->blueNoise;
    ctx->sampler.blueNoiseLayers = (scene->blueNoise && params->useBlueNoise) ? 4u : 0u;
    const V4 invSize(1.0f / (float)(int32_t)width, 1.0f / (float)(int32_t)height, 0.0f, 0.0f);
    const V4 sampleOffset(params->sampleOffset[0], params->sampleOffset[1], 0.0f, 0.0f);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x)
        {
            const uint32_t realY = height - 1u - y;
            const V4 coords = (V4((float)(int32_t)x, (float)(int32_t)realY, 0.0f, 0.0f) + sampleOffset) * invSize;
            ctx->sampler.resetPixel(x, y, params->rngKey);
            ctx->simd.resetPixel(x, y, params->rngKey);
            (void)cameraGenerateRay(params->camera, coords, ctx->sampler);   // Viewport::RenderTile generates the ray before RenderPixel ignores it
            c.c[C_PRIMARY]++;
            lightTracerPixel(*ctx);
        }
    delete ctx;
    if (counters) for (int i = 0; i < 16; ++i) counters[i] += c.c[i];
    return 0;
}

uint32_t rto_sizeof(int what)
{
    switch (what)
    {
    case 0: return (uint32_t)sizeof(RtSceneDesc);
    case 1: return (uint32_t)sizeof(RtPassParams);
    case 2: return (uint32_t)sizeof(RtObject);
    case 3: return (uint32_t)sizeof(RtLight);
    case 4: return (uint32_t)sizeof(RtMaterial);
    case 5: return (uint32_t)sizeof(RtCamera);
    case 6: return (uint32_t)sizeof(RtTexture);
    default: return 0;
    }
}

// x86 approximation mode of rto_math.h (FastDivide, fastNormalized3).  Returns 1 if the host has the instructions, 0 otherwise (the mode stays off).
// signature[0..1] (may be null): the bits of this CPU's rcp_ss(3.0) and rsqrt_ps(0.7) -- approximation tables differ between CPU families, the fixtures
// of tests/golden/ref_render were rendered on one of them (tests/test_reference_images.py skips the every-pixel comparison elsewhere).
int rto_set_x86_approximations(int enable, uint32_t* signature)
{
#ifdef RTO_HAVE_X86_APPROX
    g_rtoX86Approximations = enable ? 1 : 0;
    if (signature)
    {
        float three = 3.0f;
        const float a = _mm_cvtss_f32(_mm_rcp_ss(_mm_load_ss(&three))), b = _mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(0.7f)));
        memcpy(&signature[0], &a, 4); memcpy(&signature[1], &b, 4);
    }
    return 1;
#else
    (void)enable; if (signature) { signature[0] = 0u; signature[1] = 0u; }
    return 0;
#endif
}

} // extern "C"
