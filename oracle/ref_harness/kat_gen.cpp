// oracle/ref_harness/kat_gen.cpp -- TEST INFRASTRUCTURE, build container only.
//
// Known-answer-test generator: calls the REFERENCE's own functions (the translation units under
// /root/reference/Core that compile unmodified on Linux, see Makefile) on seeded inputs and writes
// input/output records to tests/golden/*.kat.  Nothing of the reference is copied: this file only
// #includes its public headers and links its objects.  The fixtures are data (numbers), committed so that
// the parity of the CPU oracle (oracle/rto_*.h) and of the host-side algorithms (BVH builder, Halton,
// Random) can be checked on machines where /root/reference does not exist.
//
// File format ("KAT1"): uint32 magic, funcId, n, inStride, outStride, reserved; then n*inStride float32
// inputs and n*outStride float32 outputs (integers are bit-cast into the float slots).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <memory>
#include <sstream>
#include <algorithm>
#include <functional>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <map>
#include <unordered_map>
#include <array>
#include <limits>
#include <cmath>

// test-only access to private state (generator seeds); layouts are unchanged
#define private public
#define protected public
#include "PCH.h"
#include "Math/Math.h"
#include "Math/Vector4.h"
#include "Math/Random.h"
#include "Math/Transcendental.h"
#include "Math/SamplingHelpers.h"
#include "Math/Geometry.h"
#include "Math/Utils.h"
#include "Math/Matrix4.h"
#include "Math/Transform.h"
#include "Math/Quaternion.h"
#include "Sampling/HaltonSampler.h"
#include "Sampling/GenericSampler.h"
#include "BVH/BVH.h"
#include "BVH/BVHBuilder.h"
#include "Shapes/SphereShape.h"
#include "Shapes/BoxShape.h"
#include "Shapes/RectShape.h"
#include "Shapes/MeshShape.h"
#include "Scene/Light/AreaLight.h"
#include "Scene/Light/BackgroundLight.h"
#include "Scene/Light/DirectionalLight.h"
#include "Scene/Light/PointLight.h"
#include "Scene/Light/SpotLight.h"
#include "Scene/Object/SceneObject_Shape.h"
#include "Scene/Object/SceneObject_Light.h"
#include "Scene/Camera.h"
#include "Rendering/Film.h"
#include "Math/Packed.h"
#include "Material/Material.h"
#include "Textures/BitmapTexture.h"
#include "Textures/CheckerboardTexture.h"
#include "Textures/NoiseTexture.h"
#include "Textures/MixTexture.h"
#include "Utils/Bitmap.h"
#include "Rendering/PostProcess.h"
#include "Color/ColorHelpers.h"
#include "Math/Half.h"
#include "Material/BSDF/BSDF.h"
#include "Material/BSDF/Microfacet.h"
#include "Rendering/Context.h"
#include "Rendering/ShadingData.h"
#include "Traversal/TraversalContext.h"
#include "Traversal/HitPoint.h"
#include "Traversal/Intersection.h"
#undef private
#undef protected

#include "../../include/rtgpu.h"

using namespace rt;
using namespace rt::math;

static std::string gOutDir = "tests/golden";

// ---- deterministic inputs ---------------------------------------------------------------------------
struct Lcg
{
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(seed * 2862933555777941757ULL + 3037000493ULL) {}
    uint32_t u32() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(s >> 33) ^ (uint32_t)(s >> 7); }
    float unit() { return (float)(u32() >> 8) * (1.0f / 16777216.0f); }                 // [0, 1)
    float range(float a, float b) { return a + (b - a) * unit(); }
    Vector4 vec(float a, float b) { return Vector4(range(a, b), range(a, b), range(a, b), 0.0f); }
    Vector4 dir() { Vector4 v; do { v = vec(-1.0f, 1.0f); } while (v.SqrLength3() < 0.01f || v.SqrLength3() > 1.0f); return v.Normalized3(); }
};

static float bitsf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

struct KatWriter
{
    uint32_t func, inStride, outStride;
    std::vector<float> in, out;
    std::string name;
    KatWriter(const char* n, uint32_t f, uint32_t is, uint32_t os) : func(f), inStride(is), outStride(os), name(n) {}
    float* addIn() { in.resize(in.size() + inStride, 0.0f); return in.data() + in.size() - inStride; }
    float* addOut() { out.resize(out.size() + outStride, 0.0f); return out.data() + out.size() - outStride; }
    void save()
    {
        const std::string path = gOutDir + "/" + name + ".kat";
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(1); }
        const uint32_t n = (uint32_t)(in.size() / inStride);
        const uint32_t header[6] = { 0x3154414Bu /* 'KAT1' */, func, n, inStride, outStride, 0 };
        fwrite(header, sizeof(header), 1, f);
        fwrite(in.data(), 4, in.size(), f);
        fwrite(out.data(), 4, out.size(), f);
        fclose(f);
        printf("wrote %s (%u records)\n", path.c_str(), n);
    }
};

static void put4(float* o, const Vector4& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
static void putM(float* o, const Matrix4& m) { for (int i = 0; i < 4; ++i) put4(o + 4 * i, m.rows[i]); }

// KAT function ids -- must match oracle/rt_oracle.cpp
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
    // bidirectional (VCM) building blocks
    KAT_LIGHT_EMIT = 42, KAT_LIGHT_ILLUMINATE_BIDIR = 43, KAT_LIGHT_RADIANCE_BIDIR = 44, KAT_BSDF_PDFS = 52,
    KAT_CAMERA_FILM = 61, KAT_FILM_SPLAT = 62, KAT_PACKED_PHOTON = 63, KAT_HSV_TO_RGB = 64,
    // host-side algorithms (checked against raytracer_amd's host library, not the oracle)
    KAT_HOST_EULER = 100, KAT_HOST_INVERSE = 101,
};

static Matrix4 randomRigid(Lcg& g)
{
    const Float3 euler(g.range(-3.0f, 3.0f), g.range(-3.0f, 3.0f), g.range(-3.0f, 3.0f));
    return Transform(g.vec(-5.0f, 5.0f), Quaternion::FromEulerAngles(euler)).ToMatrix4();
}

// =====================================================================================================
static void genMath()
{
    const int N = 512;
    {
        KatWriter k("math_sin_lane", KAT_SIN_LANE, 1, 1); Lcg g(1);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = g.range(-20.0f, 20.0f); k.addOut()[0] = Sin(Vector4(in[0])).x; }
        k.save();
    }
    {
        KatWriter k("math_sincos", KAT_SINCOS, 1, 4); Lcg g(2);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = g.range(0.0f, 6.2831853f); put4(k.addOut(), SinCos(in[0])); }
        k.save();
    }
    {
        KatWriter k("math_fastlog", KAT_FASTLOG, 1, 1); Lcg g(3);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = (i < N / 2) ? g.unit() * 0.999f + 1.0e-6f : g.range(0.001f, 1000.0f); k.addOut()[0] = FastLog(in[0]); }
        k.save();
    }
    {
        KatWriter k("math_fastacos", KAT_FASTACOS, 1, 1); Lcg g(4);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = g.range(-1.0f, 1.0f); k.addOut()[0] = FastACos(in[0]); }
        k.save();
    }
    {
        KatWriter k("math_fastatan2", KAT_FASTATAN2, 2, 1); Lcg g(5);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = g.range(-2.0f, 2.0f); in[1] = g.range(-2.0f, 2.0f); k.addOut()[0] = FastATan2(in[0], in[1]); }
        k.save();
    }
    {
        KatWriter k("math_float_normal2", KAT_FLOAT_NORMAL2, 2, 4); Lcg g(6);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = g.unit() * 0.999f + 1.0e-4f; in[1] = g.unit(); put4(k.addOut(), SamplingHelpers::GetFloatNormal2(Float2(in[0], in[1]))); }
        k.save();
    }
    {
        KatWriter k("math_hemisphere_cos", KAT_HEMISPHERE_COS, 2, 4); Lcg g(7);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = g.unit(); in[1] = g.unit(); put4(k.addOut(), SamplingHelpers::GetHemishpereCos(Float2(in[0], in[1]))); }
        k.save();
    }
    {
        KatWriter k("math_sphere", KAT_SPHERE, 2, 4); Lcg g(8);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = g.unit(); in[1] = g.unit(); put4(k.addOut(), SamplingHelpers::GetSphere(Float2(in[0], in[1]))); }
        k.save();
    }
    {
        KatWriter k("math_circle", KAT_CIRCLE, 2, 4); Lcg g(9);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = g.unit(); in[1] = g.unit(); put4(k.addOut(), SamplingHelpers::GetCircle(Float2(in[0], in[1]))); }
        k.save();
    }
    {
        KatWriter k("math_ortho_basis", KAT_ORTHO_BASIS, 4, 8); Lcg g(10);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); const Vector4 n = g.dir(); put4(in, n); Vector4 u, v; BuildOrthonormalBasis(n, u, v); float* o = k.addOut(); put4(o, u); put4(o + 4, v); }
        k.save();
    }
    {
        KatWriter k("math_fresnel_dielectric", KAT_FRESNEL_DIELECTRIC, 2, 1); Lcg g(11);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = g.range(-1.0f, 1.0f); in[1] = g.range(1.0f, 2.5f); k.addOut()[0] = FresnelDielectric(in[0], in[1]); }
        k.save();
    }
    {
        KatWriter k("math_fresnel_metal", KAT_FRESNEL_METAL, 3, 1); Lcg g(12);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); in[0] = g.range(0.0f, 1.0f); in[1] = g.range(0.0f, 3.0f); in[2] = g.range(0.0f, 100.0f); k.addOut()[0] = FresnelMetal(in[0], in[1], in[2]); }
        k.save();
    }
    {
        KatWriter k("math_refract3", KAT_REFRACT3, 9, 4); Lcg g(13);
        for (int i = 0; i < N; ++i)
        {
            float* in = k.addIn(); const Vector4 d = g.dir(); Vector4 n = (i & 1) ? VECTOR_Z : g.dir();
            put4(in, d); put4(in + 4, n); in[8] = g.range(1.0f, 2.0f);
            put4(k.addOut(), Vector4::Refract3(d, n, in[8]));
        }
        k.save();
    }
    {
        KatWriter k("math_reflect3", KAT_REFLECT3, 8, 4); Lcg g(14);
        for (int i = 0; i < N; ++i) { float* in = k.addIn(); const Vector4 d = g.dir(), n = g.dir(); put4(in, d); put4(in + 4, n); put4(k.addOut(), Vector4::Reflect3(d, n)); }
        k.save();
    }
}

// =====================================================================================================
static void genGeometry()
{
    const int N = 1024;
    {
        KatWriter k("geom_make_ray", KAT_MAKE_RAY, 8, 12); Lcg g(20);
        for (int i = 0; i < N; ++i)
        {
            float* in = k.addIn(); const Vector4 o = g.vec(-10.0f, 10.0f); Vector4 d = g.vec(-1.0f, 1.0f);
            if (i % 17 == 0) d.x = 0.0f;            // axis-parallel components: invDir = inf, originDivDir = inf / nan
            if (i % 29 == 0) { d.y = 0.0f; d.z = 0.0f; d.x = 1.0f; }
            put4(in, o); put4(in + 4, d);
            const Ray r(o, d);
            float* out = k.addOut(); put4(out, r.dir); put4(out + 4, r.invDir); put4(out + 8, r.originDivDir);
        }
        k.save();
    }
    {
        KatWriter k("geom_transform_ray", KAT_TRANSFORM_RAY, 24, 16); Lcg g(21);
        for (int i = 0; i < N; ++i)
        {
            float* in = k.addIn(); const Matrix4 m = randomRigid(g).Inverse();
            const Ray w(g.vec(-10.0f, 10.0f), g.dir());
            putM(in, m); put4(in + 16, w.origin); put4(in + 20, w.dir);
            const Ray l = m.TransformRay_Unsafe(w);
            float* out = k.addOut(); put4(out, l.origin); put4(out + 4, l.dir); put4(out + 8, l.invDir); put4(out + 12, l.originDivDir);
        }
        k.save();
    }
    {
        KatWriter k("geom_fast_inverse", KAT_FAST_INVERSE, 16, 16); Lcg g(22);
        for (int i = 0; i < 256; ++i) { float* in = k.addIn(); const Matrix4 m = randomRigid(g); putM(in, m); putM(k.addOut(), m.FastInverseNoScale()); }
        k.save();
    }
    {
        // Round 6 (review item 6): what Scene::Traverse_Object / EvaluateIntersection (Scene.cpp:128-165, 305-348 -- a translation unit that does not build here) call
        // on an INSTANCE'S matrices, on rotated AND scaled transforms: Matrix4::TransformPoint / TransformVector, and FastInverseNoScale followed by TransformPoint
        // (Scene.cpp:313-317: not an inverse once the matrix scales -- the reference's behaviour, pinned as it is)
        KatWriter k("geom_transform_scaled", KAT_TRANSFORM_SCALED, 20, 12); Lcg g(26);
        for (int i = 0; i < N; ++i)
        {
            Matrix4 m = randomRigid(g);
            if (i % 3 != 0) { const Vector4 scale = (i % 3 == 1) ? Vector4(g.range(0.25f, 4.0f)) : g.vec(0.25f, 4.0f); for (int r = 0; r < 3; ++r) m.rows[r] = m.rows[r] * (r == 0 ? scale.x : (r == 1 ? scale.y : scale.z)); }
            const Vector4 v = g.vec(-10.0f, 10.0f);
            float* in = k.addIn(); putM(in, m); put4(in + 16, v);
            float* out = k.addOut(); put4(out, m.TransformPoint(v)); put4(out + 4, m.TransformVector(v)); put4(out + 8, m.FastInverseNoScale().TransformPoint(v));
        }
        k.save();
    }
    {
        // The frame composition of Scene::EvaluateIntersection, Scene.cpp:311-348, from the reference's own inline functions in the reference's order: local hit
        // position, normal mapping in the tangent frame (FastNormalized3), Vector4::Orthogonalize + Normalized3 of the tangent, both vectors to world space
        // (TransformVector), Cross3 for the bitangent.  in: transform[16], ray origin[4], ray direction[4], {distance, normal-mapped?, 0, 0}, local tangent[4],
        // local normal[4], tangent-space normal of the map[4]; out: local position[4], frame rows 0..3.
        KatWriter k("frame_compose", KAT_FRAME_COMPOSE, 40, 20); Lcg g(27);
        for (int i = 0; i < 2 * N; ++i)
        {
            Matrix4 transform = randomRigid(g);
            if (i % 4 == 3) { const float scale = g.range(0.5f, 2.0f); for (int r = 0; r < 3; ++r) transform.rows[r] = transform.rows[r] * scale; }
            const Ray ray(g.vec(-10.0f, 10.0f), g.dir());
            const float distance = g.range(0.01f, 50.0f);
            const bool mapped = (i & 1) != 0;
            // interpolated vertex data: unit-ish, not exactly orthogonal
            const Vector4 normal = (g.dir() + g.vec(-0.05f, 0.05f));
            Vector4 tangent = Vector4::Cross3(normal, g.dir()).Normalized3() + g.vec(-0.1f, 0.1f) + normal * g.range(-0.2f, 0.2f);
            Vector4 mapNormal = Vector4(g.range(-0.6f, 0.6f), g.range(-0.6f, 0.6f), 0.0f, 0.0f); mapNormal.z = sqrtf(1.0f - mapNormal.x * mapNormal.x - mapNormal.y * mapNormal.y);
            float* in = k.addIn(); putM(in, transform); put4(in + 16, ray.origin); put4(in + 20, ray.dir); in[24] = distance; in[25] = mapped ? 1.0f : 0.0f;
            put4(in + 28, tangent); put4(in + 32, normal); put4(in + 36, mapNormal);

            const Matrix4 invTransform = transform.FastInverseNoScale();
            const Vector4 worldPosition = ray.GetAtDistance(distance);
            const Vector4 localPosition = invTransform.TransformPoint(worldPosition);
            Vector4 localSpaceTangent = tangent, localSpaceNormal = normal;
            const Vector4 localSpaceBitangent = Vector4::Cross3(localSpaceTangent, localSpaceNormal);
            if (mapped)
            {
                Vector4 newNormal = localSpaceTangent * mapNormal.x;
                newNormal = Vector4::MulAndAdd(localSpaceBitangent, mapNormal.y, newNormal);
                newNormal = Vector4::MulAndAdd(localSpaceNormal, mapNormal.z, newNormal);
                localSpaceNormal = newNormal.FastNormalized3();
            }
            localSpaceTangent = Vector4::Orthogonalize(localSpaceTangent, localSpaceNormal).Normalized3();
            Vector4 frame[4];
            frame[2] = transform.TransformVector(localSpaceNormal);
            frame[0] = transform.TransformVector(localSpaceTangent);
            frame[1] = Vector4::Cross3(frame[0], frame[2]);
            frame[3] = worldPosition;
            float* out = k.addOut(); put4(out, localPosition); for (int r = 0; r < 4; ++r) put4(out + 4 + 4 * r, frame[r]);
        }
        k.save();
    }
    {
        KatWriter k("geom_box_ray", KAT_BOX_RAY, 14, 2); Lcg g(23);
        KatWriter k2("geom_box_ray_twosided", KAT_BOX_RAY_TWOSIDED, 14, 3);
        for (int i = 0; i < 4 * N; ++i)
        {
            const Vector4 c = g.vec(-3.0f, 3.0f), e = g.vec(0.05f, 2.0f);
            Box box(c - e, c + e); box.min.w = 0.0f; box.max.w = 0.0f;
            Vector4 o = g.vec(-6.0f, 6.0f); Vector4 d = (i % 3 == 0) ? (c - o) + g.vec(-1.0f, 1.0f) : g.vec(-1.0f, 1.0f);
            if (i % 13 == 0) d.y = 0.0f;
            if (i % 31 == 0) { o = c; }               // origin inside the box
            if (i % 37 == 0) { o.x = box.min.x; d.x = 0.0f; }   // 0 * inf on a slab plane
            const Ray r(o, d);
            float* in = k.addIn(); put4(in, o); put4(in + 4, d); in[8] = box.min.x; in[9] = box.min.y; in[10] = box.min.z; in[11] = box.max.x; in[12] = box.max.y; in[13] = box.max.z;
            float dist = 0.0f; const bool h = Intersect_BoxRay(r, box, dist);
            float* out = k.addOut(); out[0] = bitsf(h ? 1u : 0u); out[1] = dist;
            float* in2 = k2.addIn(); memcpy(in2, in, 14 * 4);
            float nd = 0.0f, fd = 0.0f; const bool h2 = Intersect_BoxRay_TwoSided(r, box, nd, fd);
            float* out2 = k2.addOut(); out2[0] = bitsf(h2 ? 1u : 0u); out2[1] = nd; out2[2] = fd;
        }
        k.save(); k2.save();
    }
    {
        KatWriter k("geom_triangle_ray", KAT_TRIANGLE_RAY, 17, 4); Lcg g(24);
        for (int i = 0; i < 4 * N; ++i)
        {
            const Vector4 v0 = g.vec(-3.0f, 3.0f), v1 = v0 + g.vec(-1.5f, 1.5f), v2 = v0 + g.vec(-1.5f, 1.5f);
            const ProcessedTriangle tri(v0, v1, v2);
            const Vector4 o = g.vec(-6.0f, 6.0f);
            const float a = g.unit(), b = g.unit() * (1.0f - a);
            Vector4 target = v0 + (v1 - v0) * a + (v2 - v0) * b;
            if (i % 5 == 0) target = v0 + (v1 - v0) * g.range(-0.5f, 1.5f) + (v2 - v0) * g.range(-0.5f, 1.5f);
            if (i % 11 == 0) target = v0 + (v1 - v0) * a;           // exactly on an edge (strict inequalities)
            if (i % 23 == 0) target = v1;                           // exactly on a vertex
            const Vector4 d = (i % 7 == 3) ? g.vec(-1.0f, 1.0f) : target - o;
            const Ray r(o, d);
            float* in = k.addIn(); put4(in, o); put4(in + 4, d);
            in[8] = tri.v0.x; in[9] = tri.v0.y; in[10] = tri.v0.z; in[11] = tri.edge1.x; in[12] = tri.edge1.y; in[13] = tri.edge1.z; in[14] = tri.edge2.x; in[15] = tri.edge2.y; in[16] = tri.edge2.z;
            float u = 0, v = 0, t = 0;
            const bool h = Intersect_TriangleRay(r, Vector4(tri.v0), Vector4(tri.edge1), Vector4(tri.edge2), u, v, t);
            float* out = k.addOut(); out[0] = bitsf(h ? 1u : 0u); out[1] = u; out[2] = v; out[3] = t;
        }
        k.save();
    }
}

// =====================================================================================================
static std::unique_ptr<IShape> makeShape(uint32_t kind, Lcg& g, float param[4], float param2[4])
{
    for (int i = 0; i < 4; ++i) { param[i] = 0.0f; param2[i] = 0.0f; }
    if (kind == RT_SHAPE_SPHERE) { const float r = g.range(0.3f, 2.5f); param[0] = r; param[1] = 1.0f / r; return std::make_unique<SphereShape>(r); }
    if (kind == RT_SHAPE_BOX)
    {
        const Vector4 s = g.vec(0.3f, 2.5f); param[0] = s.x; param[1] = s.y; param[2] = s.z;
        const Vector4 inv = VECTOR_ONE / s; param2[0] = inv.x; param2[1] = inv.y; param2[2] = inv.z;
        return std::make_unique<BoxShape>(s);
    }
    const Float2 s(g.range(0.3f, 2.5f), g.range(0.3f, 2.5f)); const Float2 ts(g.range(0.5f, 2.0f), g.range(0.5f, 2.0f));
    param[0] = s.x; param[1] = s.y; param[2] = ts.x; param[3] = ts.y;
    return std::make_unique<RectShape>(s, ts);
}

static void genShapes()
{
    const int N = 1536;
    KatWriter ki("shape_intersect", KAT_SHAPE_INTERSECT, 13, 4), ks("shape_sample", KAT_SHAPE_SAMPLE, 12, 8),
        kp("shape_pdf", KAT_SHAPE_PDF, 13, 1), ke("shape_eval", KAT_SHAPE_EVAL, 13, 16);
    Lcg g(30);
    for (int i = 0; i < N; ++i)
    {
        const uint32_t kind = (uint32_t)(i % 3);
        float p[4], p2[4];
        std::unique_ptr<IShape> shape = makeShape(kind, g, p, p2);
        // --- Intersect
        {
            Vector4 o = g.vec(-5.0f, 5.0f); Vector4 d = (i % 4 == 0) ? g.vec(-1.0f, 1.0f) : (g.vec(-1.0f, 1.0f) * (kind == RT_SHAPE_RECT ? Vector4(1.0f, 1.0f, 0.0f, 0.0f) : VECTOR_ONE)) - o;
            if (i % 19 == 0) o = g.vec(-0.2f, 0.2f);   // origin inside
            const Ray r(o, d);
            float* in = ki.addIn(); in[0] = bitsf(kind); memcpy(in + 1, p, 16); put4(in + 5, o); put4(in + 9, d);
            ShapeIntersection si; si.nearDist = 0.0f; si.farDist = 0.0f;
            const bool h = shape->Intersect(r, si);
            float* out = ki.addOut(); out[0] = bitsf(h ? 1u : 0u); out[1] = h ? si.nearDist : 0.0f; out[2] = h ? si.farDist : 0.0f; out[3] = bitsf(si.subObjectId);
        }
        // --- Sample(ref, u)
        {
            const Vector4 ref = g.vec(-6.0f, 6.0f); const Float3 u(g.unit(), g.unit(), g.unit());
            float* in = ks.addIn(); in[0] = bitsf(kind); memcpy(in + 1, p, 16); put4(in + 5, ref); in[9] = u.x; in[10] = u.y; in[11] = u.z;
            ShapeSampleResult sr;
            const bool h = shape->Sample(ref, u, sr);
            float* out = ks.addOut(); out[0] = bitsf(h ? 1u : 0u);
            if (h) { put4(out + 1, sr.direction); out[5] = sr.distance; out[6] = sr.pdf; out[7] = sr.cosAtSurface; }
        }
        // --- Pdf(ref, point)
        {
            const Vector4 ref = g.vec(-6.0f, 6.0f); Vector4 pt = g.dir() * (kind == RT_SHAPE_SPHERE ? p[0] : 1.0f);
            float* in = kp.addIn(); in[0] = bitsf(kind); memcpy(in + 1, p, 16); put4(in + 5, ref); put4(in + 9, pt);
            kp.addOut()[0] = shape->Pdf(ref, pt);
        }
        // --- EvaluateIntersection
        {
            Vector4 pos;
            if (kind == RT_SHAPE_SPHERE) pos = g.dir() * p[0];
            else if (kind == RT_SHAPE_BOX) { pos = g.vec(-1.0f, 1.0f) * Vector4(p[0], p[1], p[2], 0.0f); const int ax = (int)(g.u32() % 3); pos[ax] = (g.u32() & 1) ? p[ax] : -p[ax]; }
            else pos = g.vec(-1.0f, 1.0f) * Vector4(p[0], p[1], 0.0f, 0.0f);
            float* in = ke.addIn(); in[0] = bitsf(kind); memcpy(in + 1, p, 16); memcpy(in + 5, p2, 16); put4(in + 9, pos);
            IntersectionData id; id.frame = Matrix4::Zero(); id.frame[3] = pos; id.texCoord = Vector4::Zero();
            HitPoint hp;
            shape->EvaluateIntersection(hp, id);
            float* out = ke.addOut(); put4(out, id.frame[0]); put4(out + 4, id.frame[1]); put4(out + 8, id.frame[2]); put4(out + 12, id.texCoord);
        }
    }
    ki.save(); ks.save(); kp.save(); ke.save();
}

// =====================================================================================================
static void fillLight(RtLight& L, const ILight& light, const Matrix4& xf, uint32_t shapeKind, const float p[4], const float p2[4])
{
    memset(&L, 0, sizeof(L));
    memcpy(L.transform, &xf, 64);
    const Matrix4 inv = xf.Inverse();
    memcpy(L.invTransform, &inv, 64);
    memcpy(L.color, &light.GetColor().rgbValues, 16);
    L.type = (uint32_t)light.GetType();
    L.flags = (uint32_t)light.GetFlags();
    L.shapeKind = shapeKind;
    L.texture = RT_NO_TEXTURE;
    memcpy(L.shapeParam, p, 16); memcpy(L.shapeParam2, p2, 16);
}

static void genLights()
{
    const int N = 1280;
    const uint32_t LW = sizeof(RtLight) / 4;
    KatWriter ki("light_illuminate", KAT_LIGHT_ILLUMINATE, LW + 16 + 3, 11), kr("light_radiance", KAT_LIGHT_RADIANCE, LW + 13, 5);
    Lcg g(40);
    RenderingContext* ctx = new RenderingContext();
    for (int i = 0; i < N; ++i)
    {
        const int type = i % 5;
        float p[4] = { 0, 0, 0, 0 }, p2[4] = { 0, 0, 0, 0 };
        uint32_t shapeKind = 0;
        const Vector4 color(g.range(0.0f, 6.0f), g.range(0.0f, 6.0f), g.range(0.0f, 6.0f), (i % 7 == 0) ? 1.0f : 0.0f);
        std::unique_ptr<ILight> light;
        float cosAngle = 0.0f; uint32_t isDelta = 0;
        if (type == 0)
        {
            shapeKind = (uint32_t)((i / 5) % 3);
            std::unique_ptr<IShape> shape = makeShape(shapeKind, g, p, p2);
            light = std::make_unique<AreaLight>(ShapePtr(std::move(shape)), color);
        }
        else if (type == 1) light = std::make_unique<BackgroundLight>(color);
        else if (type == 2)
        {
            const float angle = ((i / 5) % 3 == 0) ? 0.005f : g.range(0.02f, 0.8f);
            auto dl = std::make_unique<DirectionalLight>(color, angle);
            cosAngle = dl->mCosAngle; isDelta = dl->mIsDelta ? 1u : 0u;
            light = std::move(dl);
        }
        else if (type == 3) light = std::make_unique<PointLight>(color);
        else
        {
            const float angle = ((i / 5) % 4 == 0) ? 0.005f : g.range(0.1f, 1.4f);
            auto sl = std::make_unique<SpotLight>(color, angle);
            cosAngle = sl->mCosAngle; isDelta = sl->mIsDelta ? 1u : 0u;
            light = std::move(sl);
        }
        const Matrix4 xf = randomRigid(g);
        RtLight L; fillLight(L, *light, xf, shapeKind, p, p2);
        L.cosAngle = cosAngle; L.isDelta = isDelta;

        // --- Illuminate
        {
            IntersectionData isect;
            const Vector4 n = g.dir(); Vector4 t, b; BuildOrthonormalBasis(n, t, b);
            isect.frame[0] = t; isect.frame[1] = b; isect.frame[2] = n; isect.frame[3] = g.vec(-6.0f, 6.0f);
            const Float3 u(g.unit(), g.unit(), g.unit());
            float* in = ki.addIn(); memcpy(in, &L, sizeof(L)); putM(in + LW, isect.frame); in[LW + 16] = u.x; in[LW + 17] = u.y; in[LW + 18] = u.z;
            const ILight::IlluminateParam param = { xf.Inverse(), xf, isect, ctx->wavelength, u };
            ILight::IlluminateResult res;
            const RayColor rad = light->Illuminate(param, res);
            float* out = ki.addOut(); put4(out, rad.value);
            const bool zero = rad.AlmostZero() && type == 0;   // area light early-out leaves the result untouched
            if (!zero) { put4(out + 4, res.directionToLight); out[8] = res.distance; out[9] = res.directPdfW; out[10] = res.cosAtLight; }
            else { put4(out + 4, Vector4::Zero()); out[8] = -1.0f; out[9] = -1.0f; out[10] = -1.0f; }
        }
        // --- GetRadiance (hittable lights only: area, background, directional)
        if (type <= 2)
        {
            const Ray lray(g.vec(-6.0f, 6.0f), g.dir());
            Vector4 hit = g.vec(-2.0f, 2.0f);
            if (type == 0 && shapeKind == RT_SHAPE_SPHERE) hit = g.dir() * p[0];
            const float cosAtLight = g.range(-0.3f, 1.0f);
            float* in = kr.addIn(); memcpy(in, &L, sizeof(L)); put4(in + LW, lray.origin); put4(in + LW + 4, lray.dir); put4(in + LW + 8, hit); in[LW + 12] = cosAtLight;
            const ILight::RadianceParam param = { *ctx, lray, hit, cosAtLight };
            float pdf = 0.0f;
            const RayColor rad = light->GetRadiance(param, &pdf);
            float* out = kr.addOut(); put4(out, rad.value); out[4] = rad.AlmostZero() ? 0.0f : pdf;
        }
    }
    ki.save(); kr.save();
}

// =====================================================================================================
static const char* kBsdfNames[9] = { "null", "diffuse", "roughDiffuse", "dielectric", "roughDielectric", "metal", "roughMetal", "plastic", "roughPlastic" };

static void genBsdf()
{
    const int N = 4608;
    KatWriter ks("bsdf_sample", KAT_BSDF_SAMPLE, 23, 11), ke("bsdf_evaluate", KAT_BSDF_EVALUATE, 24, 5);
    Lcg g(50);
    Wavelength wavelength;
    for (int i = 0; i < N; ++i)
    {
        const uint32_t kind = (uint32_t)(i % 9);
        Material mat;
        mat.SetBsdf(kBsdfNames[kind]);
        mat.baseColor = Vector4(g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), (i % 11 == 0) ? 1.0f : 0.0f);
        mat.emission = Vector4::Zero();
        mat.roughness = (i % 13 == 0) ? 0.001f : g.range(0.02f, 1.0f);
        mat.metalness = 0.0f;
        mat.IoR = (kind == 5 || kind == 6) ? g.range(0.0f, 3.0f) : g.range(1.05f, 2.2f);
        mat.K = g.range(0.0f, 8.0f);
        mat.Compile();
        RtMaterial M; memset(&M, 0, sizeof(M));
        memcpy(M.emission, &mat.emission.baseValue, 16); memcpy(M.baseColor, &mat.baseColor.baseValue, 16);
        M.roughness = mat.roughness.baseValue; M.metalness = mat.metalness.baseValue; M.IoR = mat.IoR; M.K = mat.K; M.bsdf = kind;

        SampledMaterialParameters mp;
        mp.baseColor = RayColor(mat.baseColor.baseValue); mp.emissionColor = RayColor(mat.emission.baseValue);
        mp.roughness = M.roughness; mp.metalness = M.metalness; mp.IoR = M.IoR;

        Vector4 outgoing = g.dir();
        if (i % 3 != 0) outgoing.z = Abs(outgoing.z);        // mostly the upper hemisphere
        if (i % 41 == 0) outgoing = Vector4(1.0f, 0.0f, 0.0f, 0.0f);   // grazing
        // --- Sample
        {
            const Float3 u(g.unit(), g.unit(), g.unit());
            float* in = ks.addIn(); memcpy(in, &M, 64); put4(in + 16, outgoing); in[20] = u.x; in[21] = u.y; in[22] = u.z;
            BSDF::SamplingContext sc = { mat, mp, u, outgoing, wavelength };
            const bool ok = mat.GetBSDF()->Sample(sc);
            float* out = ks.addOut(); out[0] = bitsf(ok ? 1u : 0u);
            if (ok) { put4(out + 1, sc.outColor.value); put4(out + 5, sc.outIncomingDir); out[9] = sc.outPdf; out[10] = bitsf((uint32_t)sc.outEventType); }
        }
        // --- Evaluate
        {
            Vector4 incoming = g.dir();
            if (i % 4 != 0) incoming.z = -Abs(incoming.z);   // mostly arriving from above (NdotL = -incoming.z > 0)
            float* in = ke.addIn(); memcpy(in, &M, 64); put4(in + 16, outgoing); put4(in + 20, incoming);
            const BSDF::EvaluationContext ec = { mat, mp, wavelength, outgoing, incoming };
            float pdf = 0.0f;
            const RayColor c = mat.GetBSDF()->Evaluate(ec, &pdf);
            float* out = ke.addOut(); put4(out, c.value); out[4] = c.AlmostZero() ? 0.0f : pdf;
        }
    }
    ks.save(); ke.save();
}

// =====================================================================================================
static void genCamera()
{
    const int N = 512;
    const uint32_t CW = sizeof(RtCamera) / 4;
    KatWriter k("camera_ray", KAT_CAMERA_RAY, CW + 8, 16);
    Lcg g(60);
    RenderingContext* ctx = new RenderingContext();
    for (int i = 0; i < N; ++i)
    {
        Camera cam;
        const Float3 euler(g.range(-1.5f, 1.5f), g.range(-3.0f, 3.0f), g.range(-0.5f, 0.5f));
        cam.SetTransform(Transform(g.vec(-10.0f, 10.0f), Quaternion::FromEulerAngles(euler)));
        cam.SetPerspective(g.range(0.5f, 2.4f), g.range(0.2f, 2.0f));
        cam.mDOF.enable = (i % 2) == 1;
        cam.mDOF.focalPlaneDistance = g.range(0.5f, 20.0f);
        cam.mDOF.aperture = g.range(0.01f, 0.5f);
        cam.mDOF.bokehShape = (BokehShape)((i / 2) % 3);                 // circle, hexagon, square
        if (i % 5 >= 3) { cam.barrelDistortionVariableFactor = g.range(0.005f, 0.05f); cam.barrelDistortionConstFactor = g.range(0.0f, 0.02f); }
        RtCamera C; memset(&C, 0, sizeof(C));
        memcpy(C.localToWorld, &cam.mLocalToWorld, 64);
        C.aspectRatio = cam.mAspectRatio; C.tanHalfFoV = cam.mTanHalfFoV; C.dofEnable = cam.mDOF.enable ? 1u : 0u;
        C.focalPlaneDistance = cam.mDOF.focalPlaneDistance; C.aperture = cam.mDOF.aperture;
        C.bokehShape = (uint32_t)cam.mDOF.bokehShape;
        C.barrelDistortionConstFactor = cam.barrelDistortionConstFactor; C.barrelDistortionVariableFactor = cam.barrelDistortionVariableFactor;
        const Vector4 coords(g.unit(), g.unit(), 0.0f, 0.0f);
        // the two DOF dimensions: seed values with no blue noise and a zero salt chain are not available through
        // the public API, so drive the sampler state directly: mCurrentSample = {u0, u1}, no dithering, salt = 0
        const uint32_t s0 = g.u32(), s1 = g.u32();
        DynArray<uint32> seed; seed.PushBack(s0); seed.PushBack(s1);
        ctx->sampler.ResetFrame(seed, false);
        ctx->sampler.mBlueNoisePixelX = 0; ctx->sampler.mBlueNoisePixelY = 0; ctx->sampler.mSalt = 0; ctx->sampler.mSamplesGenerated = 0;
        ctx->randomGenerator.mSeed[0] = ((uint64)g.u32() << 32) | g.u32(); ctx->randomGenerator.mSeed[1] = ((uint64)g.u32() << 32) | g.u32() | 1u;
        float* in = k.addIn(); memcpy(in, &C, sizeof(C)); in[CW] = coords.x; in[CW + 1] = coords.y; in[CW + 2] = bitsf(s0); in[CW + 3] = bitsf(s1);
        memcpy(in + CW + 4, &ctx->randomGenerator.mSeed[0], 8); memcpy(in + CW + 6, &ctx->randomGenerator.mSeed[1], 8);
        const Ray r = cam.GenerateRay(coords, *ctx);
        float* out = k.addOut(); put4(out, r.origin); put4(out + 4, r.dir); put4(out + 8, r.invDir); put4(out + 12, r.originDivDir);
    }
    k.save();
}

// =====================================================================================================
// Integer generators and host-side algorithms: raw binary fixtures with their own small headers
// =====================================================================================================
static void writeRaw(const char* name, const void* data, size_t bytes)
{
    const std::string path = gOutDir + "/" + name;
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(1); }
    fwrite(data, 1, bytes, f);
    fclose(f);
    printf("wrote %s (%zu bytes)\n", path.c_str(), bytes);
}

static void genIntegers()
{
    // ---- Random: xoroshiro128+ scalar stream and the two xorshift128+ lanes behind GetVector4
    {
        Random r;
        const uint64_t scalar[2] = { 0x0123456789ABCDEFULL, 0xFEDCBA9876543210ULL };
        const uint64_t simd[4] = { 0x1111111122222222ULL, 0x3333333344444444ULL, 0x5555555566666666ULL, 0x7777777788888888ULL };
        r.mSeed[0] = scalar[0]; r.mSeed[1] = scalar[1];
        memcpy(&r.mSeedSimd4[0], &simd[0], 16); memcpy(&r.mSeedSimd4[1], &simd[2], 16);
        std::vector<uint8_t> blob;
        auto push = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; blob.insert(blob.end(), b, b + n); };
        const uint32_t count = 64;
        push(scalar, 16); push(simd, 32); push(&count, 4);
        for (uint32_t i = 0; i < count; ++i) { const uint64_t v = r.GetLong(); push(&v, 8); }
        for (uint32_t i = 0; i < count; ++i) { const Vector4 v = r.GetVector4(); push(&v, 16); }
        // GetFloat / GetDouble from a fresh copy of the scalar state
        r.mSeed[0] = scalar[0]; r.mSeed[1] = scalar[1];
        for (uint32_t i = 0; i < count; ++i) { const float v = r.GetFloat(); push(&v, 4); }
        r.mSeed[0] = scalar[0]; r.mSeed[1] = scalar[1];
        for (uint32_t i = 0; i < count; ++i) { const double v = r.GetDouble(); push(&v, 8); }
        writeRaw("random.bin", blob.data(), blob.size());
    }
    // ---- Halton: seed[64] of the first 16 passes from a fixed private-generator state, and a 128-dim run
    for (uint32_t dims : { 64u, 128u })
    {
        HaltonSequence h;
        const uint64_t scalar[2] = { 0x9E3779B97F4A7C15ULL, 0xD1B54A32D192ED03ULL };
        h.mRandom.mSeed[0] = scalar[0]; h.mRandom.mSeed[1] = scalar[1];
        h.Initialize(dims);
        const uint32_t passes = 16;
        std::vector<uint32_t> blob; blob.push_back(dims); blob.push_back(passes);
        blob.push_back((uint32_t)scalar[0]); blob.push_back((uint32_t)(scalar[0] >> 32)); blob.push_back((uint32_t)scalar[1]); blob.push_back((uint32_t)(scalar[1] >> 32));
        for (uint32_t p = 0; p < passes; ++p) { h.NextSample(); for (uint32_t d = 0; d < dims; ++d) blob.push_back(h.GetInt(d)); }
        char name[64]; snprintf(name, sizeof(name), "halton_%u.bin", dims);
        writeRaw(name, blob.data(), blob.size() * 4);
    }
    // ---- GenericSampler: GetInt for dims 0..79 (16 past the 64 Halton dims -> fallback generator) at fixed pixels
    {
        Random fallback;
        GenericSampler s;
        s.fallbackGenerator = &fallback;
        if (!s.mBlueNoiseTexture) { fprintf(stderr, "blue noise texture not loaded: run from a directory one level below Data/\n"); exit(1); }
        Lcg g(70);
        DynArray<uint32> seed; for (int i = 0; i < 64; ++i) seed.PushBack(g.u32());
        const uint32_t pixels[][2] = { { 0, 0 }, { 1, 0 }, { 127, 127 }, { 128, 5 }, { 1919, 1079 }, { 640, 479 }, { 65535, 65535 }, { 3, 40000 } };
        const uint32_t numPixels = sizeof(pixels) / sizeof(pixels[0]);
        std::vector<uint32_t> blob; blob.push_back(64); blob.push_back(numPixels); blob.push_back(64);
        for (int i = 0; i < 64; ++i) blob.push_back(seed[i]);
        for (uint32_t useBlue = 0; useBlue < 2; ++useBlue)
        {
            s.ResetFrame(seed, useBlue != 0);
            for (uint32_t p = 0; p < numPixels; ++p)
            {
                blob.push_back(pixels[p][0]); blob.push_back(pixels[p][1]);
                s.ResetPixel(pixels[p][0], pixels[p][1]);
                for (int d = 0; d < 64; ++d) blob.push_back(s.GetInt());
            }
        }
        // GetFloat of a few extreme integers
        writeRaw("sampler.bin", blob.data(), blob.size() * 4);
    }
}


// A BVH's node array as eight words per node {min.xyz, childIndex, max.xyz, numLeaves | splitAxis << 30}.  Two things the reference's builder leaves
// UNINITIALISED are written as zero so that a regenerated fixture is byte-identical to the committed one (tests/test_golden_regeneration.py):
// node 1 (the root is node 0 and child pairs start at 2: BVHBuilder.cpp never writes it) and the splitAxis bits of a leaf.
static void pushNodes(std::vector<uint32_t>& blob, const BVH& bvh)
{
    for (uint32_t i = 0; i < bvh.GetNumNodes(); ++i)
    {
        if (i == 1u) { for (int k = 0; k < 8; ++k) blob.push_back(0u); continue; }
        const BVH::Node& nd = bvh.GetNodes()[i];
        blob.push_back(fbits(nd.min.x)); blob.push_back(fbits(nd.min.y)); blob.push_back(fbits(nd.min.z)); blob.push_back(nd.childIndex);
        blob.push_back(fbits(nd.max.x)); blob.push_back(fbits(nd.max.y)); blob.push_back(fbits(nd.max.z));
        blob.push_back(nd.numLeaves | ((nd.numLeaves == 0 ? nd.splitAxis : 0u) << 30));
    }
}

static void genHost()
{
    // ---- transforms: Euler (degrees, JSON convention) -> matrix, and the general inverse
    {
        KatWriter k("host_euler", KAT_HOST_EULER, 6, 16), ki("host_inverse", KAT_HOST_INVERSE, 16, 16); Lcg g(80);
        for (int i = 0; i < 256; ++i)
        {
            float* in = k.addIn(); for (int j = 0; j < 3; ++j) in[j] = g.range(-10.0f, 10.0f);
            in[3] = (i % 4 == 0) ? 90.0f : g.range(-180.0f, 180.0f); in[4] = (i % 5 == 0) ? 180.0f : g.range(-180.0f, 180.0f); in[5] = (i % 3 == 0) ? 0.0f : g.range(-180.0f, 180.0f);
            Vector4 orientation(in[3], in[4], in[5], 0.0f); orientation *= (RT_PI / 180.0f);
            const Matrix4 m = Transform(Vector4(in[0], in[1], in[2], 0.0f), Quaternion::FromEulerAngles(orientation.ToFloat3())).ToMatrix4();
            putM(k.addOut(), m);
            putM(ki.addIn(), m); putM(ki.addOut(), m.Inverse());
        }
        k.save(); ki.save();
    }
    // ---- BVH builder: node arrays + leaf order for random box sets
    {
        std::vector<uint32_t> blob; Lcg g(81);
        const uint32_t sizes[] = { 1, 2, 3, 10, 64, 777, 5000 };
        blob.push_back((uint32_t)(sizeof(sizes) / sizeof(sizes[0])));
        for (uint32_t n : sizes)
        {
            DynArray<Box> boxes;
            for (uint32_t i = 0; i < n; ++i)
            {
                const Vector4 c = g.vec(-20.0f, 20.0f); Vector4 e = g.vec(0.0f, 1.5f);
                if (i % 9 == 0) e.y = 0.0f;   // flat boxes (axis-aligned triangles)
                Box b(c - e, c + e); b.min.w = 0.0f; b.max.w = 0.0f;
                if (n > 50 && i % 50 == 1) b = boxes[i - 1];   // exact duplicates: equal centres
                boxes.PushBack(b);
            }
            BVH bvh; BVHBuilder builder(bvh); BVHBuilder::Indices order;
            builder.Build(boxes.Data(), n, BvhBuildingParams(), order);
            blob.push_back(n); blob.push_back(bvh.GetNumNodes());
            for (uint32_t i = 0; i < n; ++i) { const float f[6] = { boxes[i].min.x, boxes[i].min.y, boxes[i].min.z, boxes[i].max.x, boxes[i].max.y, boxes[i].max.z }; for (float v : f) blob.push_back(fbits(v)); }
            pushNodes(blob, bvh);
            for (uint32_t i = 0; i < n; ++i) blob.push_back(order[i]);
        }
        writeRaw("bvh_builder.bin", blob.data(), blob.size() * 4);
    }
}

// =====================================================================================================
// Mesh path: MeshShape::Initialize on the mesh in tests/golden/mesh_input.bin (written by
// tests/golden/make_mesh_input.py), then Traverse / Traverse_Shadow / EvaluateIntersection for seeded rays.
// =====================================================================================================
static void genMesh()
{
    const std::string inPath = gOutDir + "/mesh_input.bin";
    FILE* f = fopen(inPath.c_str(), "rb");
    if (!f) { printf("skipping mesh KAT: %s not found\n", inPath.c_str()); return; }
    uint32_t hdr[3];
    if (fread(hdr, 4, 3, f) != 3) { fclose(f); return; }
    const uint32_t nv = hdr[0], nt = hdr[1], nmat = hdr[2];
    std::vector<Float3> pos(nv), nrm(nv), tan(nv); std::vector<Float2> uv(nv); std::vector<uint32_t> idx(3 * nt), mat(nt);
    bool ok = fread(pos.data(), 12, nv, f) == nv && fread(nrm.data(), 12, nv, f) == nv && fread(tan.data(), 12, nv, f) == nv &&
              fread(uv.data(), 8, nv, f) == nv && fread(idx.data(), 12, nt, f) == nt && fread(mat.data(), 4, nt, f) == nt;
    fclose(f);
    if (!ok) { fprintf(stderr, "bad mesh_input.bin\n"); exit(1); }
    std::vector<MaterialPtr> materials;
    for (uint32_t i = 0; i < nmat; ++i) { MaterialPtr m = Material::Create(); m->SetBsdf("diffuse"); m->Compile(); materials.push_back(m); }
    MeshDesc desc;
    desc.vertexBufferDesc.numVertices = nv; desc.vertexBufferDesc.numTriangles = nt; desc.vertexBufferDesc.numMaterials = nmat;
    desc.vertexBufferDesc.vertexIndexBuffer = idx.data(); desc.vertexBufferDesc.positions = pos.data(); desc.vertexBufferDesc.normals = nrm.data();
    desc.vertexBufferDesc.tangents = tan.data(); desc.vertexBufferDesc.texCoords = uv.data(); desc.vertexBufferDesc.materialIndexBuffer = mat.data();
    desc.vertexBufferDesc.materials = materials.data();
    MeshShape mesh;
    if (!mesh.Initialize(desc)) { fprintf(stderr, "MeshShape::Initialize failed\n"); exit(1); }

    std::vector<uint32_t> blob;
    // BVH nodes + triangles (leaf order) + vertex indices
    const BVH& bvh = mesh.mBVH;
    blob.push_back(bvh.GetNumNodes()); blob.push_back(nt);
    pushNodes(blob, bvh);
    for (uint32_t i = 0; i < nt; ++i)
    {
        const ProcessedTriangle& t = mesh.mVertexBuffer.GetTriangle(i);
        const float v[9] = { t.v0.x, t.v0.y, t.v0.z, t.edge1.x, t.edge1.y, t.edge1.z, t.edge2.x, t.edge2.y, t.edge2.z };
        for (float x : v) blob.push_back(fbits(x));
        VertexIndices vi; mesh.mVertexBuffer.GetVertexIndices(i, vi);
        blob.push_back(vi.i0); blob.push_back(vi.i1); blob.push_back(vi.i2); blob.push_back(vi.materialIndex);
    }
    // rays
    const Box bb = mesh.GetBoundingBox();
    const Vector4 center = bb.GetCenter(), ext = (bb.max - bb.min) * 0.5f;
    Lcg g(90);
    RenderingContext* ctx = new RenderingContext();
    const uint32_t numRays = 4096;
    blob.push_back(numRays);
    for (uint32_t i = 0; i < numRays; ++i)
    {
        const Vector4 o = center + g.vec(-1.0f, 1.0f) * ext * ((i % 3 == 0) ? 2.5f : 0.9f);
        const Vector4 target = center + g.vec(-1.0f, 1.0f) * ext;
        const Vector4 d = target - o;
        const Ray ray(o, d);
        const float tmax = (i % 2) ? g.range(0.1f, 3.0f) * ext.Length3() : std::numeric_limits<float>::infinity();
        for (int j = 0; j < 3; ++j) blob.push_back(fbits(o[j]));
        for (int j = 0; j < 3; ++j) blob.push_back(fbits(d[j]));
        blob.push_back(fbits(tmax));
        // closest hit
        HitPoint hp; hp.distance = tmax; hp.objectId = RT_INVALID_OBJECT; hp.subObjectId = 0; hp.u = 0.0f; hp.v = 0.0f;
        const SingleTraversalContext tc = { ray, hp, *ctx };
        mesh.Traverse(tc, 7);
        blob.push_back(hp.objectId); blob.push_back(hp.objectId == 7 ? hp.subObjectId : 0u); blob.push_back(fbits(hp.distance));
        blob.push_back(fbits(hp.objectId == 7 ? hp.u : 0.0f)); blob.push_back(fbits(hp.objectId == 7 ? hp.v : 0.0f));
        // any hit
        HitPoint hs; hs.distance = tmax; hs.objectId = RT_INVALID_OBJECT;
        const SingleTraversalContext ts = { ray, hs, *ctx };
        blob.push_back(mesh.Traverse_Shadow(ts) ? 1u : 0u);
        // shading frame at the hit (local space, before Scene::EvaluateIntersection's orthogonalisation)
        float fr[13]; for (float& x : fr) x = 0.0f;
        uint32_t matIndex = 0xFFFFFFFFu;
        if (hp.objectId == 7)
        {
            IntersectionData id; id.frame = Matrix4::Zero(); id.texCoord = Vector4::Zero(); id.material = nullptr;
            mesh.EvaluateIntersection(hp, id);
            fr[0] = id.frame[0].x; fr[1] = id.frame[0].y; fr[2] = id.frame[0].z; fr[3] = id.frame[0].w;
            fr[4] = id.frame[2].x; fr[5] = id.frame[2].y; fr[6] = id.frame[2].z; fr[7] = id.frame[2].w;
            fr[8] = id.texCoord.x; fr[9] = id.texCoord.y; fr[10] = id.texCoord.z; fr[11] = id.texCoord.w;
            for (uint32_t m = 0; m < nmat; ++m) if (id.material == materials[m].get()) matIndex = m;
        }
        for (int j = 0; j < 12; ++j) blob.push_back(fbits(fr[j]));
        blob.push_back(matIndex);
    }
    writeRaw("mesh_kat.bin", blob.data(), blob.size() * 4);
}


// =====================================================================================================
// Textures on the shading path: BitmapTexture::Evaluate over every format the device decodes, both colour spaces,
// the three filters; CheckerboardTexture; Material::EvaluateShadingData / GetNormalVector with textures;
// BackgroundLight with an environment map.  File "texture_kat.bin":
//   u32 magic 'TEX1', numTextures, numEval, numMaterial, numBackground; u64 texelBytes
//   RtTexture[numTextures]; texel blob;
//   eval records:       u32 texture, f32 u, v, f32 out[4]
//   material records:   RtMaterial (80 B), f32 u, v, f32 baseColor[4], emission[4], roughness, metalness, normal[4]
//   background records: RtLight, f32 dir[4], f32 color[4]
static void genTextures()
{
    Lcg g(70);
    std::vector<RtTexture> descs;
    std::vector<uint8_t> blob;
    std::vector<std::shared_ptr<ITexture>> textures;
    struct Fmt { Bitmap::Format f; uint32_t bytes; int kind; };   // kind: 0 = unorm bytes, 1 = float, 2 = half
    const Fmt fmts[] = {
        { Bitmap::Format::R8_UNorm, 1, 0 }, { Bitmap::Format::R8G8_UNorm, 2, 0 }, { Bitmap::Format::B8G8R8_UNorm, 3, 0 },
        { Bitmap::Format::B8G8R8A8_UNorm, 4, 0 }, { Bitmap::Format::R8G8B8A8_UNorm, 4, 0 }, { Bitmap::Format::R16_UNorm, 2, 0 },
        { Bitmap::Format::R16G16_UNorm, 4, 0 }, { Bitmap::Format::R16G16B16A16_UNorm, 8, 0 }, { Bitmap::Format::R32_Float, 4, 1 },
        { Bitmap::Format::R32G32_Float, 8, 1 }, { Bitmap::Format::R32G32B32_Float, 12, 1 }, { Bitmap::Format::R32G32B32A32_Float, 16, 1 },
        { Bitmap::Format::R16_Half, 2, 2 }, { Bitmap::Format::R16G16_Half, 4, 2 }, { Bitmap::Format::R16G16B16_Half, 6, 2 },
        { Bitmap::Format::R16G16B16A16_Half, 8, 2 },
        // kind 3: any bytes, 32-bit texel (R11G11B10 with exponents kept in 1..30, R9G9B9E5); 4: palette indices; 5: B5G6R5;
        // 6: block compressed (8 bytes per 4x4 block), 7: BC5 (16 bytes per block)
        { Bitmap::Format::R11G11B10_Float, 4, 3 }, { Bitmap::Format::R9G9B9E5_SharedExp, 4, 3 }, { Bitmap::Format::B8G8R8A8_UNorm_Palette, 1, 4 },
        { Bitmap::Format::B5G6R5_UNorm, 2, 5 }, { Bitmap::Format::BC1, 0, 6 }, { Bitmap::Format::BC4, 0, 6 }, { Bitmap::Format::BC5, 0, 7 } };
    const uint32_t sizes[][2] = { { 7, 5 }, { 16, 16 }, { 1, 1 }, { 33, 2 }, { 2, 19 }, { 64, 32 } };
    int combo = 0;
    for (const Fmt& fm : fmts)
        for (int variant = 0; variant < 3; ++variant, ++combo)
        {
            uint32_t w = sizes[combo % 6][0], h = sizes[combo % 6][1];
            if (fm.kind >= 6) { w = ((w + 3u) / 4u) * 4u; h = ((h + 3u) / 4u) * 4u; }   // whole 4x4 blocks
            std::vector<uint8_t> data(fm.kind == 6 ? (size_t)w * h / 2 : (fm.kind == 7 ? (size_t)w * h : (size_t)w * h * fm.bytes));
            if (fm.kind == 0 || fm.kind >= 3) for (auto& b : data) b = (uint8_t)g.u32();
            if (fm.f == Bitmap::Format::R11G11B10_Float)   // keep the three 5-bit exponents in 1..30 (no INF / NaN / denormal patterns)
            {
                uint32_t* p32 = (uint32_t*)data.data();
                for (size_t k = 0; k < data.size() / 4; ++k)
                {
                    uint32_t v = p32[k];
                    auto fix = [&](uint32_t shift) { uint32_t e = (v >> shift) & 0x1Fu; e = 1u + e % 30u; v = (v & ~(0x1Fu << shift)) | (e << shift); };
                    fix(6); fix(17); fix(27);
                    p32[k] = v;
                }
            }
            if (false) for (auto& b : data) b = (uint8_t)g.u32();
            else if (fm.kind == 1) { float* f = (float*)data.data(); for (size_t k = 0; k < data.size() / 4; ++k) f[k] = g.range(0.0f, 2.0f); }
            else { Half* hp = (Half*)data.data(); for (size_t k = 0; k < data.size() / 2; ++k) hp[k] = Half(k % 37 == 0 ? 1.0e-6f : g.range(0.0f, 2.0f)); }
            Bitmap::InitData init;
            init.width = w; init.height = h; init.format = fm.f; init.data = data.data();
            init.linearSpace = (variant != 1);
            if (fm.kind == 4) init.paletteSize = 256;
            BitmapPtr bitmap = std::make_shared<Bitmap>("kat");
            if (!bitmap->Init(init)) { fprintf(stderr, "Bitmap::Init failed\n"); exit(1); }
            if (fm.kind == 4) for (uint32_t k = 0; k < 1024; ++k) bitmap->mPalette[k] = (uint8_t)g.u32();
            auto tex = std::make_shared<BitmapTexture>(bitmap);
            tex->mFilter = (BitmapTextureFilter)((combo + variant) % 3);
            RtTexture t; memset(&t, 0, sizeof(t));
            t.kind = RT_TEXTURE_BITMAP; t.format = (uint32_t)fm.f; t.width = w; t.height = h; t.stride = bitmap->mStride;
            t.linearSpace = init.linearSpace ? 1u : 0u; t.filter = (uint32_t)tex->mFilter;
            while (blob.size() % 16) blob.push_back(0);
            t.dataOffset = blob.size();
            blob.insert(blob.end(), bitmap->mData, bitmap->mData + (size_t)bitmap->mStride * h);
            if (fm.kind == 4)
            {
                while (blob.size() % 16) blob.push_back(0);
                t.paletteOffset = blob.size();
                blob.insert(blob.end(), bitmap->mPalette, bitmap->mPalette + 1024);
            }
            descs.push_back(t); textures.push_back(tex);
        }
    for (int k = 0; k < 2; ++k)
    {
        Vector4 a(g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), 0.0f), b(g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), 1.0f);
        textures.push_back(std::make_shared<CheckerboardTexture>(a, b));
        RtTexture t; memset(&t, 0, sizeof(t)); t.kind = RT_TEXTURE_CHECKERBOARD; memcpy(t.colorA, &a, 16); memcpy(t.colorB, &b, 16);
        descs.push_back(t);
    }
    // noise (1, 3, 6 octaves) and mix textures (leaf children, then a mix of mixes)
    for (uint32_t octaves : { 1u, 3u, 6u })
    {
        Vector4 a(g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), 0.0f), b(g.range(0.0f, 2.0f), g.range(0.0f, 2.0f), g.range(0.0f, 2.0f), 1.0f);
        textures.push_back(std::make_shared<NoiseTexture>(a, b, octaves));
        RtTexture t; memset(&t, 0, sizeof(t)); t.kind = RT_TEXTURE_NOISE; t.numOctaves = octaves; memcpy(t.colorA, &a, 16); memcpy(t.colorB, &b, 16);
        descs.push_back(t);
    }
    {
        const uint32_t n = (uint32_t)descs.size();
        const uint32_t triples[3][3] = { { 3, n - 1, n - 4 }, { n - 5, 10, n - 2 }, { n, n + 1, 4 } };   // the third mixes the first two
        for (int k = 0; k < 3; ++k)
        {
            textures.push_back(std::make_shared<MixTexture>(textures[triples[k][0]], textures[triples[k][1]], textures[triples[k][2]]));
            RtTexture t; memset(&t, 0, sizeof(t)); t.kind = RT_TEXTURE_MIX; t.mixA = triples[k][0]; t.mixB = triples[k][1]; t.mixWeight = triples[k][2];
            descs.push_back(t);
        }
    }
    for (int k = 0; k < 32; ++k) blob.push_back(0);   // the reference's loads read up to 16 bytes past a texel

    std::vector<uint32_t> out;
    auto pushf = [&](float f) { out.push_back(fbits(f)); };
    auto push4 = [&](const Vector4& v) { pushf(v.x); pushf(v.y); pushf(v.z); pushf(v.w); };
    const uint32_t numTextures = (uint32_t)descs.size();
    // --- Evaluate
    const float special[] = { 0.0f, 1.0f, -1.0f, 0.5f, 2.0f, -0.0f, 0.99999994f, -1.0e-8f, 1.0e-8f, 0.25f, 0.75f, -2.5f, 3.0f };
    uint32_t numEval = 0;
    std::vector<uint32_t> evalRecords;
    for (uint32_t ti = 0; ti < numTextures; ++ti)
        for (int k = 0; k < 96; ++k)
        {
            float u = g.range(-3.0f, 3.0f), v = g.range(-3.0f, 3.0f);
            if (k < 13) { u = special[k]; v = special[(k * 5 + 3) % 13]; }
            else if (k < 26) { v = special[k - 13]; }
            else if (k < 40) { u = (float)((k - 26) % 8) / (float)(descs[ti].width ? descs[ti].width : 16u) + (k % 2 ? 0.0f : 1.0e-7f); }   // texel edges
            const Vector4 c = textures[ti]->Evaluate(Vector4(u, v, 0.0f, 0.0f));
            out.push_back(ti); pushf(u); pushf(v); push4(c);
            ++numEval;
        }
    // --- Material::EvaluateShadingData + GetNormalVector
    const uint32_t numMaterial = 256;
    Wavelength wavelength;
    for (uint32_t i = 0; i < numMaterial; ++i)
    {
        MaterialPtr mat = Material::Create();
        mat->SetBsdf("roughPlastic");
        mat->baseColor = Vector4(g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), 0.0f);
        mat->emission = Vector4(g.range(0.0f, 3.0f), g.range(0.0f, 3.0f), g.range(0.0f, 3.0f), 0.0f);
        mat->roughness = g.range(0.05f, 1.0f);
        mat->metalness = g.range(0.0f, 1.0f);
        mat->normalMapStrength = (i % 5 == 0) ? 1.0f : g.range(0.0f, 1.5f);
        RtMaterial M; memset(&M, 0, sizeof(M));
        M.baseColorTexture = M.emissionTexture = M.roughnessTexture = M.metalnessTexture = M.normalMapTexture = RT_NO_TEXTURE;
        auto pick = [&]() { return g.u32() % numTextures; };
        if (i % 2 == 0) { M.baseColorTexture = pick(); mat->baseColor.texture = textures[M.baseColorTexture]; }
        if (i % 3 == 0) { M.emissionTexture = pick(); mat->emission.texture = textures[M.emissionTexture]; }
        if (i % 4 != 1) { M.roughnessTexture = pick(); mat->roughness.texture = textures[M.roughnessTexture]; }
        if (i % 4 != 2) { M.metalnessTexture = pick(); mat->metalness.texture = textures[M.metalnessTexture]; }
        M.normalMapTexture = pick(); mat->normalMap = textures[M.normalMapTexture];
        mat->Compile();
        memcpy(M.emission, &mat->emission.baseValue, 16); memcpy(M.baseColor, &mat->baseColor.baseValue, 16);
        M.roughness = mat->roughness.baseValue; M.metalness = mat->metalness.baseValue; M.IoR = mat->IoR; M.K = mat->K; M.bsdf = RT_BSDF_ROUGH_PLASTIC;
        M.normalMapStrength = mat->normalMapStrength;
        const float u = g.range(-2.0f, 2.0f), v = g.range(-2.0f, 2.0f);
        ShadingData sd;
        sd.intersection.texCoord = Vector4(u, v, 0.0f, 0.0f);
        mat->EvaluateShadingData(wavelength, sd);
        const Vector4 n = mat->GetNormalVector(sd.intersection.texCoord);
        uint32_t mw[sizeof(M) / 4]; memcpy(mw, &M, sizeof(M));
        for (size_t k = 0; k < sizeof(M) / 4; ++k) out.push_back(mw[k]);
        pushf(u); pushf(v); push4(sd.materialParams.baseColor.value); push4(sd.materialParams.emissionColor.value);
        pushf(sd.materialParams.roughness); pushf(sd.materialParams.metalness); push4(n);
    }
    // --- BackgroundLight::GetBackgroundColor with an environment map (through GetRadiance)
    const uint32_t numBackground = 256;
    RenderingContext* ctx = new RenderingContext();
    for (uint32_t i = 0; i < numBackground; ++i)
    {
        const Vector4 color(g.range(0.0f, 4.0f), g.range(0.0f, 4.0f), g.range(0.0f, 4.0f), 0.0f);
        BackgroundLight light(color);
        const uint32_t ti = g.u32() % numTextures;
        light.mTexture = textures[ti];
        const float p[4] = { 0, 0, 0, 0 };
        RtLight L; fillLight(L, light, Matrix4::Identity(), 0, p, p);
        L.texture = ti;
        Vector4 dir = g.dir();
        if (i % 17 == 0) dir = Vector4(0.0f, 1.0f, 0.0f, 0.0f);
        if (i % 19 == 0) dir = Vector4(0.0f, 0.0f, -1.0f, 0.0f);
        const Ray ray(Vector4::Zero(), dir);
        const ILight::RadianceParam param = { *ctx, ray, Vector4::Zero(), 1.0f };
        float pdf = 0.0f;
        const RayColor rad = light.GetRadiance(param, &pdf, nullptr);
        uint32_t lw[sizeof(L) / 4]; memcpy(lw, &L, sizeof(L));
        for (size_t k = 0; k < sizeof(L) / 4; ++k) out.push_back(lw[k]);
        push4(ray.dir); push4(rad.value);
    }
    std::vector<uint8_t> file;
    auto putU32 = [&](uint32_t v) { const uint8_t* b = (const uint8_t*)&v; file.insert(file.end(), b, b + 4); };
    putU32(0x31584554u); putU32(numTextures); putU32(numEval); putU32(numMaterial); putU32(numBackground);
    putU32(0);
    const uint64_t texelBytes = blob.size();
    file.insert(file.end(), (const uint8_t*)&texelBytes, (const uint8_t*)&texelBytes + 8);
    file.insert(file.end(), (const uint8_t*)descs.data(), (const uint8_t*)descs.data() + descs.size() * sizeof(RtTexture));
    file.insert(file.end(), blob.begin(), blob.end());
    file.insert(file.end(), (const uint8_t*)out.data(), (const uint8_t*)out.data() + out.size() * 4);
    writeRaw("texture_kat.bin", file.data(), file.size());
}

// =====================================================================================================
// OBJ ingestion: the reference's helpers::LoadMesh (Demo/MeshLoader.cpp + the vendored tinyobjloader 1.4.0) on
// tests/golden/obj/fixture.obj.  File "obj_mesh_kat.bin" (uint32 words):
//   numNodes, numTriangles, numMaterials; nodes (8 words each); per triangle: 9 floats {v0, edge1, edge2},
//   i0, i1, i2, materialIndex, then 3 x {normal xyz, tangent xyz, uv} of its vertices;
//   per material: baseColor xyz, emission xyz, roughness, hasBaseColorTexture, texture width, texture height
namespace helpers
{
using MaterialsMap = std::map<std::string, rt::MaterialPtr>;
rt::MeshShapePtr LoadMesh(const std::string& filePath, MaterialsMap& outMaterials, const float scale);
}

static void genObjMesh()
{
    helpers::MaterialsMap materialsMap;
    MeshShapePtr meshPtr = helpers::LoadMesh(gOutDir + "/obj/fixture.obj", materialsMap, 1.25f);
    if (!meshPtr) { fprintf(stderr, "helpers::LoadMesh failed\n"); exit(1); }
    MeshShape& mesh = *meshPtr;
    std::vector<uint32_t> blob;
    const BVH& bvh = mesh.mBVH;
    const uint32_t nt = mesh.mVertexBuffer.GetNumTriangles();
    const uint32_t nmat = (uint32_t)mesh.mVertexBuffer.mMaterials.Size();
    blob.push_back(bvh.GetNumNodes()); blob.push_back(nt); blob.push_back(nmat);
    pushNodes(blob, bvh);
    for (uint32_t i = 0; i < nt; ++i)
    {
        const ProcessedTriangle& t = mesh.mVertexBuffer.GetTriangle(i);
        const float v[9] = { t.v0.x, t.v0.y, t.v0.z, t.edge1.x, t.edge1.y, t.edge1.z, t.edge2.x, t.edge2.y, t.edge2.z };
        for (float x : v) blob.push_back(fbits(x));
        VertexIndices vi; mesh.mVertexBuffer.GetVertexIndices(i, vi);
        blob.push_back(vi.i0); blob.push_back(vi.i1); blob.push_back(vi.i2); blob.push_back(vi.materialIndex);
        VertexShadingData a, b, c; mesh.mVertexBuffer.GetShadingData(vi, a, b, c);
        for (const VertexShadingData* s : { &a, &b, &c })
        {
            blob.push_back(fbits(s->normal.x)); blob.push_back(fbits(s->normal.y)); blob.push_back(fbits(s->normal.z));
            blob.push_back(fbits(s->tangent.x)); blob.push_back(fbits(s->tangent.y)); blob.push_back(fbits(s->tangent.z));
            blob.push_back(fbits(s->texCoord.x)); blob.push_back(fbits(s->texCoord.y));
        }
    }
    for (uint32_t i = 0; i < nmat; ++i)
    {
        const Material& m = *mesh.mVertexBuffer.mMaterials[i];
        blob.push_back(fbits(m.baseColor.baseValue.x)); blob.push_back(fbits(m.baseColor.baseValue.y)); blob.push_back(fbits(m.baseColor.baseValue.z));
        blob.push_back(fbits(m.emission.baseValue.x)); blob.push_back(fbits(m.emission.baseValue.y)); blob.push_back(fbits(m.emission.baseValue.z));
        blob.push_back(fbits(m.roughness.baseValue));
        const BitmapTexture* tex = dynamic_cast<const BitmapTexture*>(m.baseColor.texture.get());
        blob.push_back(tex ? 1u : 0u); blob.push_back(tex ? tex->mBitmap->GetWidth() : 0u); blob.push_back(tex ? tex->mBitmap->GetHeight() : 0u);
    }
    writeRaw("obj_mesh_kat.bin", blob.data(), blob.size() * 4);
}

// =====================================================================================================
// Post-processing: the per-pixel body of Viewport::PostProcessTile (Viewport.cpp:506-547; Viewport.cpp itself cannot be
// linked here) composed from the reference's own functions in the same order -- Vector4 ops, FastLog / FastExp
// (Transcendental.cpp), ToneMap (ColorHelpers.h), Vector4::ToBGR -- without bloom and without dithering.
// File "postprocess_kat.bin": u32 count; records: RtPostprocessParams, f32 raw[3], u32 bgr, f32 toneMapped[3].
static void genPostprocess()
{
    Lcg g(110);
    std::vector<uint32_t> out;
    const uint32_t N = 4096;
    out.push_back(N);
    for (uint32_t i = 0; i < N; ++i)
    {
        RtPostprocessParams P; memset(&P, 0, sizeof(P));
        PostprocessParams params;
        if (i % 4 != 0)
        {
            params.colorFilter = Vector4(g.range(0.2f, 1.5f), g.range(0.2f, 1.5f), g.range(0.2f, 1.5f), 1.0f);
            params.exposure = g.range(-3.0f, 3.0f); params.contrast = g.range(0.4f, 1.6f); params.saturation = g.range(0.0f, 1.5f);
        }
        params.tonemapper = (Tonemapper)(i % 4);
        const uint32_t numPasses = 1u + (g.u32() % 300u);
        memcpy(P.colorFilter, &params.colorFilter, 16);
        P.exposure = params.exposure; P.contrast = params.contrast; P.saturation = params.saturation;
        P.ditheringStrength = 0.0f; P.bloomFactor = 0.0f; P.tonemapper = (uint32_t)params.tonemapper; P.numPasses = numPasses;
        float scale = 1.0f;
        switch (i % 7) { case 0: scale = 0.0f; break; case 1: scale = 1.0e-4f; break; case 2: scale = 50.0f; break; case 3: scale = 1.0e4f; break; default: break; }
        const Float3 raw(g.range(0.0f, 2.0f) * scale * numPasses, g.range(0.0f, 2.0f) * scale * numPasses, g.range(0.0f, 2.0f) * scale * numPasses);

        // Viewport::PostProcessTile
        const Vector4 colorScale = params.colorFilter * powf(2.0f, params.exposure);
        const float pixelScaling = 1.0f / (float)numPasses;
        Vector4 rgbColor(raw);   // Vector4_Load_Float3_Unsafe(mSum.GetPixelRef<Float3>(x, y)): the w lane (the next pixel) never reaches ToBGR
        rgbColor *= pixelScaling;
        const float grayscale = Vector4::Dot3(rgbColor, Vector4(0.2126f, 0.7152f, 0.0722f));
        rgbColor = Vector4::Max(Vector4::Zero(), Vector4::Lerp(Vector4(grayscale), rgbColor, params.saturation));
        rgbColor = FastExp(FastLog(rgbColor) * params.contrast);
        rgbColor *= colorScale;
        const Vector4 toneMapped = ToneMap(rgbColor, params.tonemapper);
        const uint32_t bgr = toneMapped.ToBGR();

        uint32_t pw[sizeof(P) / 4]; memcpy(pw, &P, sizeof(P));
        for (uint32_t w : pw) out.push_back(w);
        out.push_back(fbits(raw.x)); out.push_back(fbits(raw.y)); out.push_back(fbits(raw.z));
        out.push_back(bgr & 0x00FFFFFFu);
        out.push_back(fbits(toneMapped.x)); out.push_back(fbits(toneMapped.y)); out.push_back(fbits(toneMapped.z));
    }
    writeRaw("postprocess_kat.bin", out.data(), out.size() * 4);
}

// =====================================================================================================
// Building blocks of the bidirectional integrator (VertexConnectionAndMerging.cpp): ILight::Emit, Illuminate /
// GetRadiance with rendererSupportsSolidAngleSampling = false and their emission pdfs, BSDF reverse pdfs and
// BSDF::Pdf, Camera::WorldToFilm / PdfW, the jittered film splat and the packed photon fields.
// =====================================================================================================
static std::unique_ptr<ILight> makeLight(int i, Lcg& g, RtLight& L, Matrix4& xf)
{
    const int type = i % 5;
    float p[4] = { 0, 0, 0, 0 }, p2[4] = { 0, 0, 0, 0 };
    uint32_t shapeKind = 0;
    const Vector4 color(g.range(0.0f, 6.0f), g.range(0.0f, 6.0f), g.range(0.0f, 6.0f), (i % 7 == 0) ? 1.0f : 0.0f);
    std::unique_ptr<ILight> light;
    float cosAngle = 0.0f; uint32_t isDelta = 0;
    if (type == 0)
    {
        shapeKind = (uint32_t)((i / 5) % 3);
        std::unique_ptr<IShape> shape = makeShape(shapeKind, g, p, p2);
        light = std::make_unique<AreaLight>(ShapePtr(std::move(shape)), color);
    }
    else if (type == 1) light = std::make_unique<BackgroundLight>(color);
    else if (type == 2)
    {
        const float angle = ((i / 5) % 3 == 0) ? 0.005f : g.range(0.02f, 0.8f);
        auto dl = std::make_unique<DirectionalLight>(color, angle);
        cosAngle = dl->mCosAngle; isDelta = dl->mIsDelta ? 1u : 0u;
        light = std::move(dl);
    }
    else if (type == 3) light = std::make_unique<PointLight>(color);
    else
    {
        const float angle = ((i / 5) % 4 == 0) ? 0.005f : g.range(0.1f, 1.4f);
        auto sl = std::make_unique<SpotLight>(color, angle);
        cosAngle = sl->mCosAngle; isDelta = sl->mIsDelta ? 1u : 0u;
        light = std::move(sl);
    }
    xf = randomRigid(g);
    fillLight(L, *light, xf, shapeKind, p, p2);
    L.cosAngle = cosAngle; L.isDelta = isDelta;
    return light;
}

static void genBidirLights()
{
    const int N = 1280;
    const uint32_t LW = sizeof(RtLight) / 4;
    KatWriter ke("light_emit", KAT_LIGHT_EMIT, LW + 5, 15), ki("light_illuminate_bidir", KAT_LIGHT_ILLUMINATE_BIDIR, LW + 16 + 3, 12),
        kr("light_radiance_bidir", KAT_LIGHT_RADIANCE_BIDIR, LW + 13, 6);
    Lcg g(120);
    RenderingContext* ctx = new RenderingContext();
    for (int i = 0; i < N; ++i)
    {
        const int type = i % 5;
        RtLight L; Matrix4 xf;
        std::unique_ptr<ILight> light = makeLight(i, g, L, xf);
        // --- Emit
        {
            const Float3 up(g.unit(), g.unit(), g.unit()); const Float2 ud(g.unit(), g.unit());
            float* in = ke.addIn(); memcpy(in, &L, sizeof(L)); in[LW] = up.x; in[LW + 1] = up.y; in[LW + 2] = up.z; in[LW + 3] = ud.x; in[LW + 4] = ud.y;
            const ILight::EmitParam param = { xf, ctx->wavelength, up, ud };
            ILight::EmitResult res; memset(&res, 0, sizeof(res));
            const RayColor c = light->Emit(param, res);
            float* out = ke.addOut(); put4(out, c.value); put4(out + 4, res.position); put4(out + 8, res.direction);
            out[12] = res.directPdfA; out[13] = res.emissionPdfW; out[14] = res.cosAtLight;
        }
        // --- Illuminate without solid-angle sampling (all five results)
        {
            IntersectionData isect;
            const Vector4 n = g.dir(); Vector4 t, b; BuildOrthonormalBasis(n, t, b);
            isect.frame[0] = t; isect.frame[1] = b; isect.frame[2] = n; isect.frame[3] = g.vec(-6.0f, 6.0f);
            const Float3 u(g.unit(), g.unit(), g.unit());
            float* in = ki.addIn(); memcpy(in, &L, sizeof(L)); putM(in + LW, isect.frame); in[LW + 16] = u.x; in[LW + 17] = u.y; in[LW + 18] = u.z;
            const ILight::IlluminateParam param = { xf.Inverse(), xf, isect, ctx->wavelength, u, false };
            ILight::IlluminateResult res;
            const RayColor rad = light->Illuminate(param, res);
            float* out = ki.addOut(); put4(out, rad.value);
            put4(out + 4, res.directionToLight); out[8] = res.distance; out[9] = res.directPdfW; out[10] = res.emissionPdfW; out[11] = res.cosAtLight;
        }
        // --- GetRadiance without solid-angle sampling + emission pdf
        if (type <= 2)
        {
            const Ray lray(g.vec(-6.0f, 6.0f), g.dir());
            const Vector4 hit = g.vec(-2.0f, 2.0f);
            const float cosAtLight = g.range(-0.3f, 1.0f);
            float* in = kr.addIn(); memcpy(in, &L, sizeof(L)); put4(in + LW, lray.origin); put4(in + LW + 4, lray.dir); put4(in + LW + 8, hit); in[LW + 12] = cosAtLight;
            const ILight::RadianceParam param = { *ctx, lray, hit, cosAtLight, false };
            float pdfA = 0.0f, pdfW = 0.0f;
            const RayColor rad = light->GetRadiance(param, &pdfA, &pdfW);
            float* out = kr.addOut(); put4(out, rad.value); out[4] = rad.AlmostZero() ? 0.0f : pdfA; out[5] = rad.AlmostZero() ? 0.0f : pdfW;
        }
    }
    ke.save(); ki.save(); kr.save();
}

static void genBidirBsdf()
{
    const int N = 4608;
    KatWriter ke("bsdf_pdfs", KAT_BSDF_PDFS, 24, 8);
    Lcg g(121);
    Wavelength wavelength;
    for (int i = 0; i < N; ++i)
    {
        const uint32_t kind = (uint32_t)(i % 9);
        Material mat;
        mat.SetBsdf(kBsdfNames[kind]);
        mat.baseColor = Vector4(g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), (i % 11 == 0) ? 1.0f : 0.0f);
        mat.emission = Vector4::Zero();
        // rough metal / rough plastic below the specular threshold leave *outReversePdfW unwritten (RoughMetalBSDF.cpp:71-75,
        // RoughPlasticBSDF.cpp:95-98): keep those two above the threshold here
        mat.roughness = (i % 13 == 0 && kind != 6 && kind != 8) ? 0.001f : g.range(0.02f, 1.0f);
        mat.metalness = 0.0f;
        mat.IoR = (kind == 5 || kind == 6) ? g.range(0.0f, 3.0f) : g.range(1.05f, 2.2f);
        mat.K = g.range(0.0f, 8.0f);
        mat.Compile();
        RtMaterial M; memset(&M, 0, sizeof(M));
        memcpy(M.emission, &mat.emission.baseValue, 16); memcpy(M.baseColor, &mat.baseColor.baseValue, 16);
        M.roughness = mat.roughness.baseValue; M.metalness = mat.metalness.baseValue; M.IoR = mat.IoR; M.K = mat.K; M.bsdf = kind;
        SampledMaterialParameters mp;
        mp.baseColor = RayColor(mat.baseColor.baseValue); mp.emissionColor = RayColor(mat.emission.baseValue);
        mp.roughness = M.roughness; mp.metalness = M.metalness; mp.IoR = M.IoR;

        Vector4 outgoing = g.dir();
        if (i % 3 != 0) outgoing.z = Abs(outgoing.z);
        Vector4 incoming = g.dir();
        if (i % 4 != 0) incoming.z = -Abs(incoming.z);
        float* in = ke.addIn(); memcpy(in, &M, 64); put4(in + 16, outgoing); put4(in + 20, incoming);
        const BSDF::EvaluationContext ec = { mat, mp, wavelength, outgoing, incoming };
        float pdf = 0.0f, rev = 0.0f;
        const RayColor c = mat.GetBSDF()->Evaluate(ec, &pdf, &rev);
        float* out = ke.addOut(); put4(out, c.value); out[4] = c.AlmostZero() ? 0.0f : pdf; out[5] = c.AlmostZero() ? 0.0f : rev;
        out[6] = mat.GetBSDF()->Pdf(ec, BSDF::ForwardPdf); out[7] = mat.GetBSDF()->Pdf(ec, BSDF::ReversePdf);
    }
    ke.save();
}

static void genBidirCameraFilm()
{
    // ---- Camera::WorldToFilm / PdfW: RtCamera, world position, direction -> visible, film coords, pdf
    {
        const int N = 1024;
        const uint32_t CW = sizeof(RtCamera) / 4;
        KatWriter k("camera_film", KAT_CAMERA_FILM, CW + 8, 6);
        Lcg g(122);
        for (int i = 0; i < N; ++i)
        {
            Camera cam;
            const Float3 euler(g.range(-1.5f, 1.5f), g.range(-3.0f, 3.0f), g.range(-0.5f, 0.5f));
            cam.SetTransform(Transform(g.vec(-10.0f, 10.0f), Quaternion::FromEulerAngles(euler)));
            cam.SetPerspective(g.range(0.5f, 2.4f), g.range(0.2f, 2.0f));
            RtCamera C; memset(&C, 0, sizeof(C));
            memcpy(C.localToWorld, &cam.mLocalToWorld, 64);
            memcpy(C.worldToScreen, &cam.mWorldToScreen, 64);
            C.aspectRatio = cam.mAspectRatio; C.tanHalfFoV = cam.mTanHalfFoV;
            // mostly points in front of the camera, inside or near the frustum
            Vector4 local = Vector4(g.range(-1.5f, 1.5f), g.range(-1.5f, 1.5f), 1.0f, 0.0f) * g.range(0.05f, 30.0f);
            if (i % 9 == 0) local.z = -local.z;
            if (i % 31 == 0) local.z = 0.005f;     // inside the near plane
            const Vector4 world = cam.mLocalToWorld.TransformPoint(local);
            const Vector4 dir = (i % 2) ? (world - cam.mLocalToWorld.GetTranslation()).Normalized3() : g.dir();
            float* in = k.addIn(); memcpy(in, &C, sizeof(C)); put4(in + CW, world); put4(in + CW + 4, dir);
            Vector4 film = Vector4::Zero();
            const bool ok = cam.WorldToFilm(world, film);
            float* out = k.addOut(); out[0] = bitsf(ok ? 1u : 0u);
            if (ok) { out[1] = film.x; out[2] = film.y; out[3] = film.z; out[4] = film.w; }
            out[5] = cam.PdfW(dir);
        }
        k.save();
    }
    // ---- Film::AccumulateColor(pos, color, random): which pixel receives the splat, and the generator state after
    {
        const int N = 2048;
        KatWriter k("film_splat", KAT_FILM_SPLAT, 4 + 8, 2 + 8);
        Lcg g(123);
        for (int i = 0; i < N; ++i)
        {
            const uint32_t w = 1 + g.u32() % 96, h = 1 + g.u32() % 64;
            Bitmap sum;
            Bitmap::InitData id; id.width = w; id.height = h; id.format = Bitmap::Format::R32G32B32_Float;
            sum.Init(id);
            memset(sum.GetData(), 0, (size_t)w * h * 12);
            Film film(sum, nullptr);
            Vector4 pos(g.range(-0.1f, 1.1f), g.range(-0.1f, 1.1f), 0.0f, 0.0f);
            if (i % 5 == 0) { pos.x = (float)(g.u32() % (w + 1)) / (float)w; pos.y = (float)(g.u32() % (h + 1)) / (float)h; }   // pixel borders
            Random rng;
            for (int s = 0; s < 2; ++s) { rng.mSeedSimd4[s] = VectorInt4((int32)g.u32(), (int32)g.u32(), (int32)g.u32(), (int32)g.u32()); }
            float* in = k.addIn(); in[0] = pos.x; in[1] = pos.y; in[2] = bitsf(w); in[3] = bitsf(h);
            memcpy(in + 4, &rng.mSeedSimd4[0], 16); memcpy(in + 8, &rng.mSeedSimd4[1], 16);
            film.AccumulateColor(pos, Vector4(1.0f, 2.0f, 3.0f, 0.0f), rng);
            uint32_t px = 0xFFFFFFFFu, py = 0xFFFFFFFFu;
            const float* data = reinterpret_cast<const float*>(sum.GetData());
            for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) if (data[3 * ((size_t)y * w + x)] != 0.0f) { px = x; py = y; }
            float* out = k.addOut(); out[0] = bitsf(px); out[1] = bitsf(py);
            memcpy(out + 2, &rng.mSeedSimd4[0], 16); memcpy(out + 6, &rng.mSeedSimd4[1], 16);
        }
        k.save();
    }
    // ---- photon fields: PackedUnitVector3 and PackedColorRgbHdr round trips (packed bits and the unpacked vector)
    {
        const int N = 2048;
        KatWriter k("packed_photon", KAT_PACKED_PHOTON, 8, 11);
        Lcg g(124);
        for (int i = 0; i < N; ++i)
        {
            Vector4 d = g.dir();
            if (i % 17 == 0) d = Vector4(0.0f, 0.0f, (i & 1) ? 1.0f : -1.0f, 0.0f);
            if (i % 19 == 0) d = Vector4((i & 1) ? 1.0f : -1.0f, 0.0f, 0.0f, 0.0f);
            Vector4 c(g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), g.range(0.0f, 1.0f), 0.0f);
            c *= (i % 3 == 0) ? g.range(0.0f, 1000.0f) : g.range(0.0f, 2.0f);
            if (i % 23 == 0) c = Vector4::Zero();
            float* in = k.addIn(); put4(in, d); put4(in + 4, c);
            PackedUnitVector3 pd; pd.FromVector(d);
            PackedColorRgbHdr pc; pc.FromVector(c);
            float* out = k.addOut();
            memcpy(out, &pd, 4); memcpy(out + 1, &pc, 8);
            put4(out + 3, pd.ToVector()); put4(out + 7, pc.ToVector());
        }
        k.save();
    }
}

// =====================================================================================================
// Bloom: Bitmap::GaussianBlur (Core/Utils/Bitmap.cpp:880-1020) as Viewport::PerformPostProcess drives it
// (Viewport.cpp:432-452): level i = GaussianBlur(copy of level i-1, sigma = 2 * 2.5^i, n = 8).  The input image is a
// deterministic function of the pixel index (reproduced by the test), only outputs are stored:
//   bloom_kat.bin = { magic, count } then per case { width, height, numLevels, lattice step, sigma0 bits }
//                   followed, per level, by the 64-bit sum of all float bits and the pixels of the lattice (x % step == 0, y % step == 0)
// =====================================================================================================
static float bloomInput(uint32_t i)   // same integer recipe as tests/test_bloom.py
{
    uint32_t h = i * 2654435761u + 0x9E3779B9u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float base = (float)(h >> 8) * (1.0f / 16777216.0f);
    return (h & 0x3Fu) == 0u ? base * 400.0f : base * 2.0f;    // a few very bright pixels, like fireflies / light sources
}

static void genBloom()
{
    std::vector<uint32_t> out;
    struct Case { uint32_t w, h, levels, step; float sigma0; };
    const Case cases[] = { { 64, 48, 2, 1, 2.0f }, { 100, 36, 1, 1, 1.3f }, { 272, 208, 5, 4, 2.0f } };
    out.push_back(0x4D4F4C42u); out.push_back((uint32_t)(sizeof(cases) / sizeof(cases[0])));
    for (const Case& c : cases)
    {
        Bitmap img("bloom");
        Bitmap::InitData id; id.width = c.w; id.height = c.h; id.format = Bitmap::Format::R32G32B32_Float;
        if (!img.Init(id)) { fprintf(stderr, "Bitmap::Init failed\n"); exit(1); }
        float* data = reinterpret_cast<float*>(img.GetData());
        for (uint32_t i = 0; i < c.w * c.h * 3; ++i) data[i] = bloomInput(i);
        out.push_back(c.w); out.push_back(c.h); out.push_back(c.levels); out.push_back(c.step); out.push_back(fbits(c.sigma0));
        float blurSigma = c.sigma0;
        for (uint32_t level = 0; level < c.levels; ++level)
        {
            if (!img.GaussianBlur(blurSigma, 8)) { fprintf(stderr, "GaussianBlur failed\n"); exit(1); }
            blurSigma *= 2.5f;
            uint64_t sum = 0;
            for (uint32_t i = 0; i < c.w * c.h * 3; ++i) sum += fbits(data[i]);
            out.push_back((uint32_t)sum); out.push_back((uint32_t)(sum >> 32));
            for (uint32_t y = 0; y < c.h; y += c.step)
                for (uint32_t x = 0; x < c.w; x += c.step)
                    for (int k = 0; k < 3; ++k) out.push_back(fbits(data[3 * (y * c.w + x) + k]));
        }
    }
    writeRaw("bloom_kat.bin", out.data(), out.size() * 4);
}

// HSVtoRGB (Core/Color/ColorHelpers.h:133-156) as DebugRenderer's TriangleID mode drives it (DebugRenderer.cpp:98-106):
// in = { objectId, subObjectId } -> hash -> hue, saturation -> rgb
static void genDebug()
{
    const int N = 2048;
    KatWriter k("debug_triangle_id", KAT_HSV_TO_RGB, 2, 4);
    Lcg g(130);
    for (int i = 0; i < N; ++i)
    {
        const uint32_t objectId = (i % 5 == 0) ? (uint32_t)(i / 5) : g.u32() % 64u, subObjectId = (i % 3 == 0) ? g.u32() : g.u32() % 300000u;
        float* in = k.addIn(); in[0] = bitsf(objectId); in[1] = bitsf(subObjectId);
        const uint64 hash = Hash((uint64)objectId | ((uint64)subObjectId << 32));
        const float hue = (float)(uint32)hash / (float)UINT32_MAX;
        const float saturation = 0.5f + 0.5f * (float)(uint32)(hash >> 32) / (float)UINT32_MAX;
        put4(k.addOut(), HSVtoRGB(hue, saturation, 1.0f));
    }
    k.save();
}

// Bitmap::Load on the DDS fixtures (tests/golden/dds/*.dds, own data): what Bitmap::LoadDDS (Core/Utils/BitmapDDS.cpp) makes of each
// header variant.  dds_kat.bin = { count } then per file (sorted by name): { nameHash, ok, format, linearSpace, width, height, dataBytes, byteSum }
#include <dirent.h>
#include <algorithm>
static void genDds()
{
    const std::string dir = gOutDir + "/dds";
    std::vector<std::string> names;
    if (DIR* d = opendir(dir.c_str()))
    {
        while (dirent* e = readdir(d)) { const std::string n = e->d_name; if (n.size() > 4 && n.substr(n.size() - 4) == ".dds") names.push_back(n); }
        closedir(d);
    }
    std::sort(names.begin(), names.end());
    std::vector<uint32_t> out;
    out.push_back((uint32_t)names.size());
    for (const std::string& n : names)
    {
        uint32_t nameHash = 2166136261u; for (char ch : n) { nameHash ^= (uint8_t)ch; nameHash *= 16777619u; }
        Bitmap bitmap("dds");
        const bool ok = bitmap.Load((dir + "/" + n).c_str());
        out.push_back(nameHash); out.push_back(ok ? 1u : 0u);
        if (ok)
        {
            const size_t bytes = (size_t)bitmap.GetHeight() * bitmap.GetStride();
            uint32_t sum = 0; const uint8_t* data = reinterpret_cast<const uint8_t*>(bitmap.GetData());
            for (size_t i = 0; i < bytes; ++i) sum = sum * 31u + data[i];
            out.push_back((uint32_t)bitmap.GetFormat()); out.push_back(bitmap.mLinearSpace ? 1u : 0u); out.push_back(bitmap.GetWidth()); out.push_back(bitmap.GetHeight());
            out.push_back((uint32_t)bytes); out.push_back(sum);
        }
        else for (int k = 0; k < 6; ++k) out.push_back(0u);
    }
    writeRaw("dds_kat.bin", out.data(), out.size() * 4);
}

// Bitmap::Load on the BMP fixtures (tests/golden/bmp/*.bmp, own data): what Bitmap::LoadBMP (Core/Utils/BitmapBMP.cpp) makes of each.
// bmp_kat.bin = { count } then per file (sorted by name): { nameHash, ok, format, linearSpace, width, height, dataBytes, byteSum,
// paletteSize, paletteSum }
static void genBmp()
{
    const std::string dir = gOutDir + "/bmp";
    std::vector<std::string> names;
    if (DIR* d = opendir(dir.c_str()))
    {
        while (dirent* e = readdir(d)) { const std::string n = e->d_name; if (n.size() > 4 && n.substr(n.size() - 4) == ".bmp") names.push_back(n); }
        closedir(d);
    }
    std::sort(names.begin(), names.end());
    std::vector<uint32_t> out;
    out.push_back((uint32_t)names.size());
    for (const std::string& n : names)
    {
        uint32_t nameHash = 2166136261u; for (char ch : n) { nameHash ^= (uint8_t)ch; nameHash *= 16777619u; }
        Bitmap bitmap("bmp");
        const bool ok = bitmap.Load((dir + "/" + n).c_str());
        out.push_back(nameHash); out.push_back(ok ? 1u : 0u);
        if (ok)
        {
            const size_t bytes = (size_t)bitmap.GetHeight() * bitmap.GetStride();
            uint32_t sum = 0; const uint8_t* data = reinterpret_cast<const uint8_t*>(bitmap.GetData());
            for (size_t i = 0; i < bytes; ++i) sum = sum * 31u + data[i];
            uint32_t paletteSum = 0;
            for (size_t i = 0; i < (size_t)bitmap.mPaletteSize * 4u; ++i) paletteSum = paletteSum * 31u + bitmap.mPalette[i];
            out.push_back((uint32_t)bitmap.GetFormat()); out.push_back(bitmap.mLinearSpace ? 1u : 0u); out.push_back(bitmap.GetWidth()); out.push_back(bitmap.GetHeight());
            out.push_back((uint32_t)bytes); out.push_back(sum); out.push_back(bitmap.mPaletteSize); out.push_back(paletteSum);
        }
        else for (int k = 0; k < 8; ++k) out.push_back(0u);
    }
    writeRaw("bmp_kat.bin", out.data(), out.size() * 4);
}

int main(int argc, char** argv)
{
    if (argc > 1) gOutDir = argv[1];
    SetFlushDenormalsToZero(false);
    genMath();
    genGeometry();
    genShapes();
    genLights();
    genBsdf();
    genCamera();
    genIntegers();
    genHost();
    genMesh();
    genTextures();
    genObjMesh();
    genPostprocess();
    genBidirLights();
    genBidirBsdf();
    genBidirCameraFilm();
    genBloom();
    genDebug();
    genDds();
    genBmp();
    printf("done\n");
    return 0;
}
