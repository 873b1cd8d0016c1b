// rt_device_core.h -- device-side building blocks of the MI355X path tracer: per-pixel sampler, analytic
// shapes, lights, BSDFs, hit-frame evaluation and the two-level BVH traversal.  All functions are
// __device__ __forceinline__; the wavefront kernels that call them are in rt_trace.hip / rt_shade.hip / rt_tail.hip.
// File:line citations are relative to the reference repository (Witek902/Raytracer).
#pragma once

#include "rt_device_math.h"
#include "../../include/rtgpu.h"

namespace rtd {

// =====================================================================================================
// Sampler -- GenericSampler (Core/Sampling/GenericSampler.cpp:69-113, .h:29-42) + per-pixel fallback RNG
// =====================================================================================================
struct Sampler
{
    const uint32_t* seed; uint32_t numDims; uint32_t blueNoiseLayers; const uint16_t* blueNoise;
    uint32_t bx, by, salt, generated;
    Xoroshiro fallback;   // per-pixel stream (see RtPassParams::rngKey)

    __device__ __forceinline__ void resetPixel(uint32_t x, uint32_t y, uint64_t rngKey0, uint64_t rngKey1)
    {
        bx = x & 127u; by = y & 127u;                                   // :77-78
        salt = (uint32_t)murmurFmix64((uint64_t)(x | (y << 16)));       // :79
        generated = 0;                                                  // :80
        const uint64_t pix = (uint64_t)x | ((uint64_t)y << 32);
        fallback.s[0] = murmurFmix64(rngKey0 ^ pix);
        fallback.s[1] = murmurFmix64(rngKey1 + 0x9E3779B97F4A7C15ULL * (pix + 1)) | 1ULL;
    }
    __device__ __forceinline__ uint32_t fallbackInt() { return (uint32_t)xoroshiroNext(fallback); }   // Random::GetInt Random.cpp:49-52
    __device__ __forceinline__ uint32_t getInt()                        // GenericSampler.cpp:83-113
    {
        uint32_t sample;
        if (generated < numDims)
        {
            sample = seed[generated];
            if (generated < blueNoiseLayers)
            {
                const uint32_t pixelIndex = 128u * by + bx;
                sample += (uint32_t)blueNoise[4u * pixelIndex + generated] << 16;
            }
            else
            {
                const uint32_t s = salt;
                salt = xorShift32(s);
                sample += s;
            }
            generated++;
        }
        else
        {
            sample = fallbackInt();
        }
        return sample;
    }
    __device__ __forceinline__ float getFloat() { return Min(0.999999940395f, (float)getInt() / 4294967296.0f); }   // GenericSampler.h:29-32
};

// =====================================================================================================
// Hit record / intersection data
// =====================================================================================================
struct Hit { uint32_t objectId, subObjectId; float distance, u, v; };   // Core/Traversal/HitPoint.h:14-51
struct Intersection { M4 frame; V4 texCoord; uint32_t material; };      // Core/Traversal/Intersection.h:8-13

RT_DEV V4 worldToLocal(const Intersection& in, V4 w)   // Intersection.h:20-31
{
    V4 X = in.frame.r[0], Y = in.frame.r[1], Z = in.frame.r[2];
    transpose3(X, Y, Z);
    V4 r = X * w.x;
    r = mulAdd(Y, w.y, r);
    r = mulAdd(Z, w.z, r);
    return r;
}
RT_DEV V4 localToWorld(const Intersection& in, V4 l) { return transformVector(in.frame, l); }  // :15-18

#define RT_NUM_COUNTERS 12
struct Counters { uint32_t c[RT_NUM_COUNTERS]; };
enum { C_RAYS = 0, C_SHADOW = 1, C_SHADOW_HIT = 2, C_PRIMARY = 3, C_BOX = 4, C_BOX_PASS = 5, C_TRI = 6, C_TRI_PASS = 7, C_MESH_HITS = 8, C_ANALYTIC_HITS = 9, C_BOX_SHADOW = 10, C_TRI_SHADOW = 11 };

// =====================================================================================================
// Analytic shapes -- Core/Shapes/{SphereShape,BoxShape,RectShape,Shape}.cpp
// =====================================================================================================
struct ShapeHit { float nearDist, farDist; uint32_t subObjectId; };

RT_DEV bool shapeIntersect(uint32_t kind, const float* p, const Ray& ray, ShapeHit& out)
{
    out.subObjectId = 0xFFFFFFFFu;   // ShapeIntersection default, Shape.h:24-29
    if (kind == RT_SHAPE_SPHERE)     // SphereShape.cpp:29-46 (double precision island)
    {
        const double radiusD = (double)p[0];
        const double v = (double)dot3(ray.dir, neg(ray.origin));
        const double det = radiusD * radiusD - (double)sqrLength3(ray.origin) + v * v;
        if (det <= 0.0) return false;
        const double sqrtDet = sqrt(det);
        out.nearDist = (float)(v - sqrtDet);
        out.farDist = (float)(v + sqrtDet);
        out.subObjectId = 0;
        return out.farDist > out.nearDist;
    }
    if (kind == RT_SHAPE_BOX)        // BoxShape.cpp:118-125
    {
        const V4 size(p[0], p[1], p[2], 0.0f);
        out.subObjectId = 0;
        return intersectBoxRayTwoSided(ray, neg(size), size, out.nearDist, out.farDist);
    }
    // RT_SHAPE_RECT, RectShape.cpp:32-49
    const float t = -ray.origin.z * ray.invDir.z;
    if (t > FLT_EPSILON)
    {
        const V4 pos = rayAt(ray, t);
        if (absf(pos.x) < p[0] && absf(pos.y) < p[1])
        {
            out.nearDist = t; out.farDist = t;
            return true;
        }
    }
    return false;
}

RT_DEV float shapeSurfaceArea(uint32_t kind, const float* p)
{
    if (kind == RT_SHAPE_SPHERE) return 4.0f * RTD_PI * Sqr(p[0]);                       // SphereShape.cpp:24-27
    if (kind == RT_SHAPE_BOX) return 8.0f * (p[0] * (p[1] + p[2]) + p[1] * p[2]);        // BoxShape.cpp:113-116
    return 4.0f * p[0] * p[1];                                                           // RectShape.cpp:27-30
}

// IShape::Sample(u, &normal)  (area sampling): BoxShape.cpp:127-179, RectShape.cpp:51-64
RT_DEV V4 shapeSampleArea(uint32_t kind, const float* p, const float u[3], V4& outNormal)
{
    if (kind == RT_SHAPE_SPHERE)   // SphereShape.cpp:47-63 (bidirectional integrator only)
    {
        const V4 point = getSphere(u[0], u[1]);
        outNormal = point;
        return point * p[0];
    }
    if (kind == RT_SHAPE_RECT)
    {
        outNormal = V4(0.0f, 0.0f, 1.0f, 0.0f);
        // Vector4(mSize) * (2.0f * Vector4(Float2(u)) - VECTOR_ONE)
        const V4 size(p[0], p[1], 0.0f, 0.0f);
        return size * ((2.0f * V4(u[0], u[1], 0.0f, 0.0f)) - splat(1.0f));
    }
    // RT_SHAPE_BOX
    const float sx = p[0], sy = p[1], sz = p[2];
    const float cdfx = sy * sz;                 // BoxShape.cpp:103-107
    const float cdfy = cdfx + sz * sx;
    const float cdfz = cdfy + sx * sy;
    float v = u[2];
    uint32_t zAxis;
    v *= cdfz;
    if (v < cdfx) { v /= cdfx; zAxis = 0; }
    else if (v < cdfy) { v = (v - cdfx) / (cdfy - cdfx); zAxis = 1; }
    else { v = (v - cdfy) / (cdfz - cdfy); zAxis = 2; }
    // axes: x = (z + 1) % 3, y = (z + 2) % 3 ; spelled out per case to keep everything in registers
    const float nz = v < 0.5f ? -1.0f : 1.0f;
    const float a = 2.0f * u[0] - 1.0f, b = 2.0f * u[1] - 1.0f;
    V4 normal = zero4();
    V4 pos = zero4();
    if (zAxis == 0) { normal.x = nz; pos.y = a * sy; pos.z = b * sz; pos.x = nz * sx; }
    else if (zAxis == 1) { normal.y = nz; pos.z = a * sz; pos.x = b * sx; pos.y = nz * sy; }
    else { normal.z = nz; pos.x = a * sx; pos.y = b * sy; pos.z = nz * sz; }
    outNormal = normal;
    return pos;
}

struct ShapeSample { V4 direction; float distance, pdf, cosAtSurface; };

// IShape::Sample(ref, u, result): solid-angle sampling as seen from `ref` (light space)
RT_DEV bool shapeSampleFrom(uint32_t kind, const float* p, V4 ref, const float u[3], ShapeSample& r)
{
    if (kind == RT_SHAPE_SPHERE)   // SphereShape.cpp:65-108 (cone sampling)
    {
        const float radius = p[0];
        const V4 centerDir = neg(ref);
        const float centerDistSqr = sqrLength3(centerDir);
        const float centerDist = sqrtf(centerDistSqr);
        if (centerDistSqr < Sqr(radius)) return false;
        const float phi = RTD_2PI * u[1];
        const V4 sinCosPhi = sinCos(phi);
        float sinThetaMaxSqr = Sqr(radius) / centerDistSqr;
        float cosThetaMax = sqrtf(1.0f - Clamp(sinThetaMaxSqr, 0.0f, 1.0f));
        float cosTheta = Lerp(cosThetaMax, 1.0f, u[0]);
        float sinThetaSqr = 1.0f - Sqr(cosTheta);
        float sinTheta = sqrtf(sinThetaSqr);
        const V4 w = centerDir / centerDist;
        V4 tangent, bitangent;
        buildOrthonormalBasis(w, tangent, bitangent);
        r.direction = (tangent * sinCosPhi.y + bitangent * sinCosPhi.x) * sinTheta + w * cosTheta;
        r.direction = normalized3(r.direction);
        r.distance = centerDist * cosTheta - sqrtf(Max(0.0f, Sqr(radius) - centerDistSqr * sinThetaSqr));
        r.cosAtSurface = cosTheta;
        if (cosThetaMax > 0.999999f) r.pdf = FLT_MAX; else r.pdf = sphereCapPdf(cosThetaMax);
        return true;
    }
    // IShape::Sample generic (rect with solid-angle sampling off RectShape.cpp:16,66-94; box), Shape.cpp:65-91
    V4 normal;
    const V4 position = shapeSampleArea(kind, p, u, normal);
    V4 dir = ref - position;
    const float sqrDistance = sqrLength3(dir);
    if (sqrDistance > Sqr(FLT_EPSILON))
    {
        const float distance = sqrtf(sqrDistance);
        dir = dir / distance;
        const float cosNormalDir = dot3(normal, dir);
        if (cosNormalDir > FLT_EPSILON)
        {
            const float invArea = 1.0f / shapeSurfaceArea(kind, p);
            r.pdf = invArea * sqrDistance / cosNormalDir;
            r.distance = distance;
            r.cosAtSurface = cosNormalDir;
            r.direction = neg(dir);
            return true;
        }
    }
    return false;
}

// IShape::Pdf(ref, point): Shape.cpp:93-99, SphereShape.cpp:110-125
RT_DEV float shapePdf(uint32_t kind, const float* p, V4 ref, V4 point)
{
    if (kind == RT_SHAPE_SPHERE)
    {
        const float radius = p[0];
        const V4 rayDir = normalized3(point - ref);
        const V4 centerDir = neg(ref);
        const float centerDistSqr = sqrLength3(centerDir);
        const V4 normal = normalized3(point);
        const float cosAtLight = Max(0.0f, dot3(neg(rayDir), normal));
        const float sinThetaMaxSqr = Clamp(Sqr(radius) / centerDistSqr, 0.0f, 1.0f);
        const float cosThetaMax = sqrtf(1.0f - sinThetaMaxSqr);
        const float pdfW = sphereCapPdf(cosThetaMax);
        return pdfW * cosAtLight / sqrLength3(point - ref);
    }
    return 1.0f / shapeSurfaceArea(kind, p);
}

// face frames of the box, BoxShape.cpp:15-23 (rows 0..2 of the six 4x4 matrices), selected by face index
RT_DEV void boxFaceFrame(int side, V4& r0, V4& r1, V4& r2)
{
    switch (side)
    {
    case 0: r0 = V4(0, 0, 1, 0); r1 = V4(0, 1, 0, 0); r2 = V4(-1, 0, 0, 0); break;
    case 1: r0 = V4(0, 0, -1, 0); r1 = V4(0, 1, 0, 0); r2 = V4(1, 0, 0, 0); break;
    case 2: r0 = V4(1, 0, 0, 0); r1 = V4(0, 0, 1, 0); r2 = V4(0, -1, 0, 0); break;
    case 3: r0 = V4(1, 0, 0, 0); r1 = V4(0, 0, -1, 0); r2 = V4(0, 1, 0, 0); break;
    case 4: r0 = V4(-1, 0, 0, 0); r1 = V4(0, 1, 0, 0); r2 = V4(0, 0, -1, 0); break;
    default: r0 = V4(1, 0, 0, 0); r1 = V4(0, 1, 0, 0); r2 = V4(0, 0, 1, 0); break;
    }
}

// IShape::EvaluateIntersection for analytic shapes; frame[3] holds the local-space hit position on entry
RT_DEV void shapeEvaluateIntersection(uint32_t kind, const float* p, const float* p2, Intersection& out)
{
    const V4 pos = out.frame.r[3];
    if (kind == RT_SHAPE_SPHERE)     // SphereShape.cpp:159-175
    {
        out.texCoord = cartesianToSpherical(neg(pos));
        out.frame.r[2] = pos * p[1];   // * mInvRadius
        const V4 n = out.frame.r[2];
        // (n.Swizzle<2,0,0,0>() & mask<1,0,1,0>).ChangeSign<1,0,0,0>()  ==  [-n.z, 0, n.x, 0]
        out.frame.r[0] = V4(-n.z, 0.0f, n.x, 0.0f);
        out.frame.r[1] = neg(cross3(out.frame.r[0], out.frame.r[2]));
        out.frame.r[0] = fastNormalized3(out.frame.r[0]);
        out.frame.r[1] = fastNormalized3(out.frame.r[1]);
        out.frame.r[2] = fastNormalized3(out.frame.r[2]);
        return;
    }
    if (kind == RT_SHAPE_BOX)        // BoxShape.cpp:181-191 + ConvertXYZtoCubeUV :25-85
    {
        const V4 q = pos * V4(p2[0], p2[1], p2[2], 0.0f);
        const V4 a = abs4(q);
        const int isXPositive = q.x > 0 ? 1 : 0, isYPositive = q.y > 0 ? 1 : 0, isZPositive = q.z > 0 ? 1 : 0;
        float maxAxis, uc, vc; int side;
        if (a.x >= a.y && a.x >= a.z) { uc = isXPositive ? -q.z : q.z; side = isXPositive; maxAxis = a.x; vc = q.y; }
        else if (a.y >= a.x && a.y >= a.z) { vc = isYPositive ? -q.z : q.z; side = isYPositive + 2; maxAxis = a.y; uc = q.x; }
        else { uc = isZPositive ? q.x : -q.x; side = isZPositive + 4; maxAxis = a.z; vc = q.y; }
        out.texCoord = V4(uc, vc, 0.0f, 0.0f) / (2.0f * maxAxis) + splat(0.5f);
        boxFaceFrame(side, out.frame.r[0], out.frame.r[1], out.frame.r[2]);
        return;
    }
    // RT_SHAPE_RECT, RectShape.cpp:124-132
    out.texCoord = V4(pos.x, pos.y, 0.0f, 0.0f) * V4(p[2], p[3], 0.0f, 0.0f);
    out.frame.r[0] = V4(1, 0, 0, 0); out.frame.r[1] = V4(0, 1, 0, 0); out.frame.r[2] = V4(0, 0, 1, 0);
}

// =====================================================================================================
// Scene traversal -- Core/Scene/Scene.cpp:128-261, Core/Traversal/Traversal_Single.h, MeshShape.cpp
//
// Same algorithm and visiting order as the reference (near child first, far child pushed; any-hit for
// shadow rays without ordering), so ties between equal-distance hits resolve identically.  The "nodes to
// visit" stacks hold node indices; their capacity is checked against the BVH depth at upload time.
// =====================================================================================================
#define RT_TOP_STACK_SIZE 32
#define RT_MESH_STACK_SIZE 64

struct NodePair { float4 a0, a1, b0, b1; };   // two adjacent 32-byte nodes = 64 contiguous bytes

RT_DEV NodePair loadNodePair(const RtNode* nodes, uint32_t firstChild)
{
    const float4* p = reinterpret_cast<const float4*>(nodes + firstChild);
    NodePair n;
    n.a0 = p[0]; n.a1 = p[1]; n.b0 = p[2]; n.b1 = p[3];
    return n;
}
RT_DEV uint32_t leavesOf(uint32_t leavesWord) { return leavesWord & 0x3FFFFFFFu; }

// Triangle fetch: 36 bytes, 4-byte aligned
RT_DEV void loadTriangle(const RtTriangle* t, V4& v0, V4& e1, V4& e2)
{
    const float* f = reinterpret_cast<const float*>(t);
    v0 = V4(f[0], f[1], f[2], 0.0f); e1 = V4(f[3], f[4], f[5], 0.0f); e2 = V4(f[6], f[7], f[8], 0.0f);
}

// Slab test with hardware min/max (v_min_f32 / v_max_f32 / v_max3 / v_min3).  Valid ONLY for rays whose invDir and
// originDivDir are finite: then fma(box, invDir, -originDivDir) can overflow to +-inf but never be NaN, and for
// NaN-free inputs IEEE minNum/maxNum equal the _mm_min_ps/_mm_max_ps selects of intersectBoxRay bit for bit
// (up to the sign of a zero, which no comparison downstream can observe).
RT_DEV bool intersectBoxRayNoNaN(const Ray& ray, float minx, float miny, float minz, float maxx, float maxy, float maxz, float& outDistance)
{
    const float ax = __fmaf_rn(minx, ray.invDir.x, -ray.originDivDir.x), bx = __fmaf_rn(maxx, ray.invDir.x, -ray.originDivDir.x);
    const float ay = __fmaf_rn(miny, ray.invDir.y, -ray.originDivDir.y), by = __fmaf_rn(maxy, ray.invDir.y, -ray.originDivDir.y);
    const float az = __fmaf_rn(minz, ray.invDir.z, -ray.originDivDir.z), bz = __fmaf_rn(maxz, ray.invDir.z, -ray.originDivDir.z);
    const float nearD = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float farD = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    outDistance = nearD;
    return (farD >= nearD) && (farD >= 0.0f);
}
// A ray with a direction component of exactly zero keeps that coordinate for its whole length, and the reference's slab test says nothing about it:
// box * inf - origin * inf is NaN whenever box plane and origin have the same sign, and _mm_min_ps / _mm_max_ps then drop the axis -- the ray is tested
// against the OTHER two slabs only and walks every node of the scene's whole column there (a next-event ray towards a light whose direction lies in a
// coordinate plane, from a wall: tens of thousands of nodes; with sponza.json's light orientation [80, 0, 0] 360 such rays per pass, 2 ms per re-trace
// launch, profiles/r05_degenerate_axis_rays.txt).  No triangle of a box that is clearly off the ray's fixed coordinate can be hit -- a node's box bounds
// its triangles' vertices exactly, and Moeller-Trumbore only accepts points inside the triangle -- so such boxes are skipped: the walk visits a SUBSET of
// the reference's nodes in the reference's order, every leaf that holds a hit is still visited, the accepted hits and their order are the reference's.
// "Clearly off": by more than 2^-7 of the magnitudes involved (>= the box's extent in that axis), five decimal orders above the rounding of the tests.
// Not applied when the intersection counters are on (they count the reference's own box tests).
// ASSUMPTIONS, stated (round-5 advisor): (1) every node's box bounds the stored float vertices of the triangles below it -- true of the reference's builder, which
// unions the triangles' own float boxes upwards (BVHBuilder.cpp: leaf boxes from the vertices, interior boxes by Box(a, b)), and of any tree uploaded through the
// reference's API; a hand-made RtSceneDesc whose boxes do not bound their triangles is outside the contract for this walk AND the reference's; (2) a point
// Moeller-Trumbore accepts lies within rounding distance of the triangle, i.e. well inside the 2^-7 margin even for slivers.  Held by
// test_axis_parallel_next_event_rays and, on 6000 near-degenerate slivers under axis-parallel suns, test_axis_parallel_rays_among_sliver_triangles: the pruned walk
// (counters off) and the unpruned one (counters on) both return the oracle's image and counters.
RT_DEV bool boxNearDegenerateAxes(const Ray& r, float minx, float miny, float minz, float maxx, float maxy, float maxz)
{
    const uint32_t inf = 0x7f800000u;
    bool near = true;
    if ((__float_as_uint(r.invDir.x) & 0x7fffffffu) == inf) { const float m = (fabsf(r.origin.x) + fabsf(minx) + fabsf(maxx)) * 0.0078125f; near = near && r.origin.x >= minx - m && r.origin.x <= maxx + m; }
    if ((__float_as_uint(r.invDir.y) & 0x7fffffffu) == inf) { const float m = (fabsf(r.origin.y) + fabsf(miny) + fabsf(maxy)) * 0.0078125f; near = near && r.origin.y >= miny - m && r.origin.y <= maxy + m; }
    if ((__float_as_uint(r.invDir.z) & 0x7fffffffu) == inf) { const float m = (fabsf(r.origin.z) + fabsf(minz) + fabsf(maxz)) * 0.0078125f; near = near && r.origin.z >= minz - m && r.origin.z <= maxz + m; }
    return near;
}
RT_DEV bool rayIsNaNFree(const Ray& r)
{
    const uint32_t inf = 0x7f800000u;
    return ((__float_as_uint(r.invDir.x) & inf) != inf) && ((__float_as_uint(r.invDir.y) & inf) != inf) && ((__float_as_uint(r.invDir.z) & inf) != inf) &&
           ((__float_as_uint(r.originDivDir.x) & inf) != inf) && ((__float_as_uint(r.originDivDir.y) & inf) != inf) && ((__float_as_uint(r.originDivDir.z) & inf) != inf);
}

// (the traversal loops themselves are the per-lane state machine of rt_device_traverse.h)

// ILight::TestRayHit: AreaLight.cpp:43-53 (nearDist, may be negative); Point/Spot never hit (PointLight.cpp:35, SpotLight.cpp:41)
RT_DEV bool lightTestRayHit(const RtLight& L, const Ray& ray, float& outDistance)
{
    if (L.type != RT_LIGHT_AREA) return false;
    ShapeHit sh;
    if (shapeIntersect(L.shapeKind, L.shapeParam, ray, sh)) { outDistance = sh.nearDist; return true; }
    return false;
}

// MeshShape::EvaluateIntersection, MeshShape.cpp:283-328
// The device copy of the scene holds the shading data DE-INDEXED: one 128-byte record per triangle (its three VertexShadingData and
// its material), in triangle order, where the C ABI's vertexIndices[] would be (vertexShading[] is not uploaded).  A hit then costs
// one memory round trip to one aligned 128-byte line instead of the index record followed by three scattered 32-byte vertices.
struct __attribute__((aligned(16))) TriangleShading { RtVertexShading v[3]; uint32_t materialIndex; uint32_t _pad[7]; };
static_assert(sizeof(TriangleShading) == 128, "TriangleShading");
RT_DEV void meshEvaluateIntersection(const RtSceneDesc& d, const RtMesh& mesh, const Hit& hit, Intersection& out)
{
    const TriangleShading& tri = reinterpret_cast<const TriangleShading*>(d.vertexIndices)[mesh.firstTriangle + hit.subObjectId];
    if (tri.materialIndex != RT_NO_MATERIAL) out.material = tri.materialIndex;
    const RtVertexShading& a = tri.v[0]; const RtVertexShading& b = tri.v[1]; const RtVertexShading& c = tri.v[2];
    const V4 coeff1 = splat(hit.u), coeff2 = splat(hit.v);
    const V4 coeff0 = splat(1.0f) - (coeff1 + coeff2);
    V4 texCoord = coeff1 * V4(b.texCoord[0], b.texCoord[1], 0.0f, 0.0f);
    texCoord = mulAdd(coeff2, V4(c.texCoord[0], c.texCoord[1], 0.0f, 0.0f), texCoord);
    texCoord = mulAdd(coeff0, V4(a.texCoord[0], a.texCoord[1], 0.0f, 0.0f), texCoord);
    out.texCoord = texCoord;
    V4 tangent = coeff1 * load3(b.tangent);
    tangent = mulAdd(coeff2, load3(c.tangent), tangent);
    tangent = mulAdd(coeff0, load3(a.tangent), tangent);
    out.frame.r[0] = fastNormalized3(tangent);
    V4 normal = coeff1 * load3(b.normal);
    normal = mulAdd(coeff2, load3(c.normal), normal);
    normal = mulAdd(coeff0, load3(a.normal), normal);
    out.frame.r[2] = normalized3(normal);
}

// =====================================================================================================
// Textures -- Core/Textures/BitmapTexture.cpp, CheckerboardTexture.cpp, Core/Utils/Bitmap.cpp, Core/Color/ColorHelpers.h
// =====================================================================================================
// Half::ToFloat (Core/Math/HalfImpl.h:69-107) is the exact half -> float conversion
RT_DEV float halfToFloat(uint16_t value) { _Float16 h; __builtin_memcpy(&h, &value, 2); return (float)h; }   // v_cvt_f32_f16
// Convert_sRGB_To_Linear on all four lanes, ColorHelpers.h:15-27
RT_DEV V4 srgbToLinear(V4 c)
{
    V4 r = mulAdd(c, splat(0.305306011f), splat(0.682171111f));
    r = mulAdd(c, r, splat(0.012522878f));
    return r * c;
}
RT_DEV uint16_t rd16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
RT_DEV float rdf(const uint8_t* p) { float v; __builtin_memcpy(&v, p, 4); return v; }
// helper::DecodeBC_Grayscale, Utils/BlockCompression.cpp:81-109
RT_DEV float bcGrayscale(const uint8_t* blockData, uint32_t x, uint32_t y)
{
    const float intColor0 = (float)blockData[0], intColor1 = (float)blockData[1];
    const float color0 = intColor0 / 255.0f, color1 = intColor1 / 255.0f;
    uint64_t code; __builtin_memcpy(&code, blockData + 2, 8);
    const uint32_t index = (uint32_t)((code >> (uint64_t)(3u * (4u * y + x))) % 8u);
    if (index == 0u) return color0;
    if (index == 1u) return color1;
    if (intColor0 > intColor1) return (color0 * (float)(8u - index) + color1 * (float)(index - 1u)) / 7.0f;
    if (index == 6u) return 0.0f;
    if (index == 7u) return 1.0f;
    return (color0 * (float)(6u - index) + color1 * (float)(index - 1u)) / 5.0f;
}
// One texel as Bitmap::GetPixel / GetPixelBlock decode it (Bitmap.cpp:335-517, 520-832; loads: Math/Vector4Load.h):
// every UNorm load is float(integer) * RN(1 / max) -- the SSE paths scale by powers of two around that, exactly.
// kSimple: the scene holds only RT_FORMAT_IS_SIMPLE bitmaps (scene class 4, below) -- the same decode, three-and-a-half cases instead of 23, small
// enough to be inlined at every call site of the shading kernel.
#define RT_FORMAT_IS_SIMPLE(f) ((f) == RT_FORMAT_B8G8R8_UNORM || (f) == RT_FORMAT_B8G8R8A8_UNORM || (f) == RT_FORMAT_R8G8B8A8_UNORM || (f) == RT_FORMAT_R16G16B16A16_HALF)
template <bool kSimple = false>
RT_DEV V4 bitmapTexel(const RtTexture& t, const uint8_t* texels, uint32_t x, uint32_t y)
{
    const uint8_t* row = texels + t.dataOffset + (size_t)t.stride * y;
    const float s8 = 1.0f / 255.0f, s16 = 1.0f / 65535.0f;
    V4 c = zero4();
    if (kSimple)
    {
        if (t.format == RT_FORMAT_R16G16B16A16_HALF) c = V4(halfToFloat(rd16(row + 8 * x)), halfToFloat(rd16(row + 8 * x + 2)), halfToFloat(rd16(row + 8 * x + 4)), halfToFloat(rd16(row + 8 * x + 6)));
        else
        {
            const uint32_t bytes = t.format == RT_FORMAT_B8G8R8_UNORM ? 3u : 4u;
            const uint8_t* e = row + bytes * x;
            const float e0 = (float)(int32_t)e[0] * s8, e1 = (float)(int32_t)e[1] * s8, e2 = (float)(int32_t)e[2] * s8;
            const float e3 = bytes == 4u ? (float)(int32_t)e[3] * s8 : 0.0f;
            c = t.format == RT_FORMAT_R8G8B8A8_UNORM ? V4(e0, e1, e2, e3) : V4(e2, e1, e0, e3);
        }
        if (!t.linearSpace) c = srgbToLinear(c);
        return c;
    }
    switch (t.format)
    {
    case RT_FORMAT_R8_UNORM:       c = splat((float)(int32_t)row[x] * s8); break;
    case RT_FORMAT_R8G8_UNORM:     c = V4((float)(int32_t)row[2 * x] * s8, (float)(int32_t)row[2 * x + 1] * s8, 0.0f, 0.0f); break;
    case RT_FORMAT_B8G8R8_UNORM:   c = V4((float)(int32_t)row[3 * x + 2] * s8, (float)(int32_t)row[3 * x + 1] * s8, (float)(int32_t)row[3 * x] * s8, 0.0f); break;
    case RT_FORMAT_B8G8R8A8_UNORM: c = V4((float)(int32_t)row[4 * x + 2] * s8, (float)(int32_t)row[4 * x + 1] * s8, (float)(int32_t)row[4 * x] * s8, (float)(int32_t)row[4 * x + 3] * s8); break;
    case RT_FORMAT_R8G8B8A8_UNORM: c = V4((float)(int32_t)row[4 * x] * s8, (float)(int32_t)row[4 * x + 1] * s8, (float)(int32_t)row[4 * x + 2] * s8, (float)(int32_t)row[4 * x + 3] * s8); break;
    case RT_FORMAT_R16_UNORM:      c = splat((float)(int32_t)rd16(row + 2 * x) * s16); break;
    case RT_FORMAT_R16G16_UNORM:   c = V4((float)(int32_t)rd16(row + 4 * x) * s16, (float)(int32_t)rd16(row + 4 * x + 2) * s16, 0.0f, 0.0f); break;
    case RT_FORMAT_R16G16B16A16_UNORM:
        c = V4((float)(int32_t)rd16(row + 8 * x) * s16, (float)(int32_t)rd16(row + 8 * x + 2) * s16, (float)(int32_t)rd16(row + 8 * x + 4) * s16, (float)(int32_t)rd16(row + 8 * x + 6) * s16); break;
    case RT_FORMAT_R32_FLOAT:          c = splat(rdf(row + 4 * x)); break;
    case RT_FORMAT_R32G32_FLOAT:       c = V4(rdf(row + 8 * x), rdf(row + 8 * x + 4), 0.0f, 0.0f); break;
    case RT_FORMAT_R32G32B32_FLOAT:    c = V4(rdf(row + 12 * x), rdf(row + 12 * x + 4), rdf(row + 12 * x + 8), 0.0f); break;
    case RT_FORMAT_R32G32B32A32_FLOAT: c = V4(rdf(row + 16 * x), rdf(row + 16 * x + 4), rdf(row + 16 * x + 8), rdf(row + 16 * x + 12)); break;
    case RT_FORMAT_R16_HALF:           c = splat(halfToFloat(rd16(row + 2 * x))); break;
    case RT_FORMAT_R16G16_HALF:        c = V4(halfToFloat(rd16(row + 4 * x)), halfToFloat(rd16(row + 4 * x + 2)), 0.0f, 0.0f); break;
    case RT_FORMAT_R16G16B16_HALF:     c = V4(halfToFloat(rd16(row + 6 * x)), halfToFloat(rd16(row + 6 * x + 2)), halfToFloat(rd16(row + 6 * x + 4)), 0.0f); break;
    case RT_FORMAT_R16G16B16A16_HALF:  c = V4(halfToFloat(rd16(row + 8 * x)), halfToFloat(rd16(row + 8 * x + 2)), halfToFloat(rd16(row + 8 * x + 4)), halfToFloat(rd16(row + 8 * x + 6))); break;
    case RT_FORMAT_B8G8R8A8_UNORM_PALETTE:   // palette entries are B8G8R8A8, Bitmap.cpp:380-386
    {
        const uint8_t* e = texels + t.paletteOffset + 4u * (size_t)row[x];
        c = V4((float)(int32_t)e[2] * s8, (float)(int32_t)e[1] * s8, (float)(int32_t)e[0] * s8, (float)(int32_t)e[3] * s8); break;
    }
    case RT_FORMAT_B5G6R5_UNORM:             // Vector4_Load_B5G6R5_Norm, Vector4Load.h:52-58
    {
        const uint32_t v = rd16(row + 2 * x);
        c = V4((float)(int32_t)(v >> 11) * (1.0f / 31.0f), (float)(int32_t)((v >> 5) & 0x3Fu) * (1.0f / 63.0f), (float)(int32_t)(v & 0x1Fu) * (1.0f / 31.0f), 0.0f); break;
    }
    case RT_FORMAT_R11G11B10_FLOAT:          // PackedFloat3::ToVector, Math/Packed.h:160-167 (bit-casts, no INF / NaN / denormal handling)
    {
        uint32_t v; __builtin_memcpy(&v, row + 4 * x, 4);
        const uint32_t xm = v & 0x3Fu, xe = (v >> 6) & 0x1Fu, ym = (v >> 11) & 0x3Fu, ye = (v >> 17) & 0x1Fu, zm = (v >> 22) & 0x1Fu, ze = (v >> 27) & 0x1Fu;
        const uint32_t bx = ((xe + 112u) << 23) | (xm << 17), by = ((ye + 112u) << 23) | (ym << 17), bz = ((ze + 112u) << 23) | (zm << 17);
        float fx, fy, fz; __builtin_memcpy(&fx, &bx, 4); __builtin_memcpy(&fy, &by, 4); __builtin_memcpy(&fz, &bz, 4);
        c = V4(fx, fy, fz, 0.0f); break;
    }
    case RT_FORMAT_R9G9B9E5_SHAREDEXP:       // SharedExpFloat3::ToVector, Math/Packed.h:126-131
    {
        uint32_t v; __builtin_memcpy(&v, row + 4 * x, 4);
        const uint32_t sb = 0x33800000u + ((v >> 27) << 23);
        float scale; __builtin_memcpy(&scale, &sb, 4);
        c = V4(scale * (float)(int32_t)(v & 0x1FFu), scale * (float)(int32_t)((v >> 9) & 0x1FFu), scale * (float)(int32_t)((v >> 18) & 0x1FFu), scale * 0.0f); break;
    }
    case RT_FORMAT_BC1:                      // DecodeBC1, Utils/BlockCompression.cpp:49-76
    {
        const uint8_t* block = texels + t.dataOffset + 8u * ((size_t)(t.width / 4u) * (y / 4u) + (x / 4u));
        const uint32_t c0 = rd16(block), c1 = rd16(block + 2);
        uint32_t code; __builtin_memcpy(&code, block + 4, 4);
        const uint32_t index = (code >> (2u * (4u * (y % 4u) + (x % 4u)))) % 4u;
        const float w = index == 0u ? 0.0f : (index == 1u ? 1.0f : (index == 2u ? 1.0f / 3.0f : 2.0f / 3.0f));
        // base colours stay in their 5/6/5 bit positions until the final scale; Lerp = MulAndAdd(v2 - v1, w, v1)
        const float r0 = (float)(int32_t)(c0 & 0xF800u), g0 = (float)(int32_t)(c0 & 0x07E0u), b0 = (float)(int32_t)(c0 & 0x001Fu);
        const float r1 = (float)(int32_t)(c1 & 0xF800u), g1 = (float)(int32_t)(c1 & 0x07E0u), b1 = (float)(int32_t)(c1 & 0x001Fu);
        c = V4(__fmaf_rn(r1 - r0, w, r0) * (1.0f / 2048.0f / 31.0f), __fmaf_rn(g1 - g0, w, g0) * (1.0f / 32.0f / 63.0f), __fmaf_rn(b1 - b0, w, b0) * (1.0f / 31.0f), __fmaf_rn(0.0f, w, 0.0f) * 0.0f); break;
    }
    case RT_FORMAT_BC4:                      // DecodeBC4, BlockCompression.cpp:113-126
    {
        const float v = bcGrayscale(texels + t.dataOffset + 8u * ((size_t)(t.width / 4u) * (y / 4u) + (x / 4u)), x % 4u, y % 4u);
        c = V4(v, v, v, 1.0f); break;
    }
    case RT_FORMAT_BC5:                      // DecodeBC5, BlockCompression.cpp:128-146 (green first, as the reference returns it)
    {
        const uint8_t* block = texels + t.dataOffset + 16u * ((size_t)(t.width / 4u) * (y / 4u) + (x / 4u));
        const float red = bcGrayscale(block, x % 4u, y % 4u), green = bcGrayscale(block + 8, x % 4u, y % 4u);
        c = V4(green, red, 0.0f, 1.0f); break;
    }
    default: break;
    }
    if (!t.linearSpace) c = srgbToLinear(c);
    return c;
}
RT_DEV float smoothStep(float x) { return x * x * (3.0f - x * 2.0f); }   // Math.h:176-179
// Vector4::Lerp(v1, v2, w) = MulAndAdd(v2 - v1, w, v1), Vector4Impl.h:58-61
RT_DEV V4 lerp4(V4 v1, V4 v2, V4 w) { return mulAdd(v2 - v1, w, v1); }
// BitmapTexture::Evaluate, BitmapTexture.cpp:32-93
template <bool kSimple = false>
RT_DEV V4 bitmapTextureEvaluate(const RtTexture& t, const uint8_t* texels, V4 coords)
{
    const int32_t sw = (int32_t)t.width, sh = (int32_t)t.height;
    const float wx = coords.x - floorf(coords.x), wy = coords.y - floorf(coords.y);   // Vector4::Mod1
    const float scx = wx * (float)sw, scy = wy * (float)sh;                             // * mFloatSize
    const float fx = floorf(scx), fy = floorf(scy);
    const int32_t ix = __float2int_rn(fx), iy = __float2int_rn(fy);                     // VectorInt4::Convert (cvtps2dq)
    int32_t tx = ix, ty = iy;
    if (!(ix < sw)) tx -= sw;                                                           // texelCoords -= AndNot(intCoords < size, size)
    if (!(iy < sh)) ty -= sh;
    if (ix < 0) tx += sw;                                                               // texelCoords += size & (intCoords < 0)
    if (iy < 0) ty += sh;
    if (t.filter == RT_FILTER_NEAREST) return bitmapTexel<kSimple>(t, texels, (uint32_t)tx, (uint32_t)ty);
    int32_t tz = tx + 1, tw = ty + 1;
    if (!(tz < sw)) tz -= sw;                                                           // wrap secondary coordinates
    if (!(tw < sh)) tw -= sh;
    // GetPixelBlock: colors[0] = (x, y), [1] = (z, y), [2] = (x, w), [3] = (z, w)
    const V4 c0 = bitmapTexel<kSimple>(t, texels, (uint32_t)tx, (uint32_t)ty), c1 = bitmapTexel<kSimple>(t, texels, (uint32_t)tz, (uint32_t)ty);
    const V4 c2 = bitmapTexel<kSimple>(t, texels, (uint32_t)tx, (uint32_t)tw), c3 = bitmapTexel<kSimple>(t, texels, (uint32_t)tz, (uint32_t)tw);
    float weightX = scx - (float)ix, weightY = scy - (float)iy;                         // scaledCoords - intCoords.ConvertToFloat()
    if (t.filter == RT_FILTER_BILINEAR_SMOOTHSTEP) { weightX = smoothStep(weightX); weightY = smoothStep(weightY); }
    const V4 value0 = lerp4(c0, c2, splat(weightY));
    const V4 value1 = lerp4(c1, c3, splat(weightY));
    return lerp4(value0, value1, splat(weightX));
}
// NoiseTexture (Core/Textures/NoiseTexture.cpp): 2D simplex noise after github.com/SRombauts/SimplexNoise; the table is
// Ken Perlin's reference permutation
__device__ static const uint8_t kNoisePermutation[256] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148,
    247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175,
    74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54,
    65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64,
    52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213,
    119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104,
    218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157,
    184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180 };
RT_DEV int32_t noiseHash(int32_t i) { return kNoisePermutation[(uint8_t)i]; }
RT_DEV float noiseGradient(int32_t hash, float x, float y)   // NoiseTexture.cpp:34-40
{
    const int32_t h = hash & 0x3F;
    const float u = h < 4 ? x : y, v = h < 4 ? y : x;
    return ((h & 1) ? -u : u) + ((h & 2) ? -2.0f * v : 2.0f * v);
}
RT_DEV int32_t floorInt(float fp) { const int32_t i = (int32_t)fp; return fp < (float)i ? (i - 1) : i; }   // Math.h:100-104
RT_DEV float noiseEvaluateInternal(float cx, float cy)   // NoiseTexture::EvaluateInternal, :58-147
{
    const float F2 = 0.366025403f, G2 = 0.211324865f;
    const float s = (cx + cy) * F2;
    const float xs = cx + s, ys = cy + s;
    const int32_t i = floorInt(xs), j = floorInt(ys);
    const float t = (float)(i + j) * G2;
    const float X0 = (float)i - t, Y0 = (float)j - t;
    const float x0 = cx - X0, y0 = cy - Y0;
    const int32_t i1 = x0 > y0 ? 1 : 0, j1 = x0 > y0 ? 0 : 1;
    const float x1 = x0 - (float)i1 + G2, y1 = y0 - (float)j1 + G2;
    const float x2 = x0 - 1.0f + 2.0f * G2, y2 = y0 - 1.0f + 2.0f * G2;
    const int32_t gi0 = noiseHash(i + noiseHash(j)), gi1 = noiseHash(i + i1 + noiseHash(j + j1)), gi2 = noiseHash(i + 1 + noiseHash(j + 1));
    float n0, n1, n2;
    float t0 = 0.5f - x0 * x0 - y0 * y0;
    if (t0 < 0.0f) n0 = 0.0f; else { t0 *= t0; n0 = t0 * t0 * noiseGradient(gi0, x0, y0); }
    float t1 = 0.5f - x1 * x1 - y1 * y1;
    if (t1 < 0.0f) n1 = 0.0f; else { t1 *= t1; n1 = t1 * t1 * noiseGradient(gi1, x1, y1); }
    float t2 = 0.5f - x2 * x2 - y2 * y2;
    if (t2 < 0.0f) n2 = 0.0f; else { t2 *= t2; n2 = t2 * t2 * noiseGradient(gi2, x2, y2); }
    return Clamp(0.5f + 22.615325f * (n0 + n1 + n2), 0.0f, 1.0f);
}
// every ITexture but MixTexture.  NOT inlined: the mix nesting and the call sites of the shade kernel would otherwise carry
// ~50 copies of the 23-way texel decode
__device__ __noinline__ static V4 textureEvaluateLeaf(const RtSceneDesc& d, const RtTexture& t, V4 coords)
{
    if (t.kind == RT_TEXTURE_CHECKERBOARD)      // CheckerboardTexture.cpp:31-40
    {
        const float wx = coords.x - floorf(coords.x), wy = coords.y - floorf(coords.y);
        const bool cond = (wx > 0.5f) != (wy > 0.5f);
        return cond ? load4(t.colorA) : load4(t.colorB);
    }
    if (t.kind == RT_TEXTURE_CONST) return load4(t.colorA);
    if (t.kind == RT_TEXTURE_NOISE)             // NoiseTexture::Evaluate, NoiseTexture.cpp:149-164
    {
        float value = 0.0f, octaveValueScale = 0.5f, octaveCoordScale = 1.0f;
        for (uint32_t i = 0; i < t.numOctaves; ++i)
        {
            value += octaveValueScale * noiseEvaluateInternal(coords.x * octaveCoordScale, coords.y * octaveCoordScale);
            octaveValueScale *= 0.5f; octaveCoordScale *= 2.0f;
        }
        return lerp4(load4(t.colorA), load4(t.colorB), splat(value));
    }
    return bitmapTextureEvaluate(t, d.texelData, coords);
}
// MixTexture::Evaluate (MixTexture.cpp:23-30) = Lerp(A(uv), B(uv), weight(uv)), all four lanes.  Mixes nest one level deep.
RT_DEV V4 textureEvaluateInner(const RtSceneDesc& d, uint32_t index, V4 coords)
{
    const RtTexture& t = d.textures[index];
    if (t.kind != RT_TEXTURE_MIX) return textureEvaluateLeaf(d, t, coords);
    return lerp4(textureEvaluateLeaf(d, d.textures[t.mixA], coords), textureEvaluateLeaf(d, d.textures[t.mixB], coords),
                 textureEvaluateLeaf(d, d.textures[t.mixWeight], coords));
}
// ITexture::Evaluate dispatch
RT_DEV V4 textureEvaluate(const RtSceneDesc& d, uint32_t index, V4 coords)
{
    const RtTexture& t = d.textures[index];
    if (t.kind != RT_TEXTURE_MIX) return textureEvaluateLeaf(d, t, coords);
    return lerp4(textureEvaluateInner(d, t.mixA, coords), textureEvaluateInner(d, t.mixB, coords), textureEvaluateInner(d, t.mixWeight, coords));
}
// the shading kernels' texture lookups: scene class 4 holds nothing but simple bitmaps, whose evaluation is inlined (no call: the callee's register
// budget is what keeps the other textured classes at two waves per SIMD)
#define RT_SIMPLE_TEXTURES(k) ((k) == 4)
template <int kLean>
RT_DEV V4 textureEvaluateK(const RtSceneDesc& d, uint32_t index, V4 coords)
{
    if (RT_SIMPLE_TEXTURES(kLean)) return bitmapTextureEvaluate<true>(d.textures[index], d.texelData, coords);
    return textureEvaluate(d, index, coords);
}
// Material::GetNormalVector, Material.cpp:120-138 (normalMap != NULL)
template <int kLean>
RT_DEV V4 materialGetNormalVector(const RtSceneDesc& d, const RtMaterial& mat, V4 uv)
{
    V4 normal = textureEvaluateK<kLean>(d, mat.normalMapTexture, uv);
    normal = mulSub(normal, 2.0f, splat(1.0f));                    // UnipolarToBipolar
    normal.z = sqrtf(Max(0.0f, 1.0f - dot2(normal, normal)));      // reconstruct Z
    return lerp4(V4(0.0f, 0.0f, 1.0f, 0.0f), normal, splat(mat.normalMapStrength));
}


// The tail of Scene::EvaluateIntersection, Scene.cpp:322-348: normal mapping in the local tangent frame, the tangent made orthogonal to the (mapped) normal, both
// taken to world space.  Held against the reference on its own by frame_compose.kat (rtgpu_kat, tests/test_gpu_kat.py).
RT_DEV void composeShadingFrame(const M4& transform, V4 worldPosition, V4 localSpaceTangent, V4 localSpaceNormal, bool mapped, V4 localNormal, M4& frame)
{
    if (mapped)
    {
        const V4 localSpaceBitangent = cross3(localSpaceTangent, localSpaceNormal);
        V4 newNormal = localSpaceTangent * localNormal.x;
        newNormal = mulAdd(localSpaceBitangent, localNormal.y, newNormal);
        newNormal = mulAdd(localSpaceNormal, localNormal.z, newNormal);
        localSpaceNormal = fastNormalized3(newNormal);
    }
    localSpaceTangent = normalized3(orthogonalize(localSpaceTangent, localSpaceNormal));   // :342
    frame.r[2] = transformVector(transform, localSpaceNormal);
    frame.r[0] = transformVector(transform, localSpaceTangent);
    frame.r[1] = cross3(frame.r[0], frame.r[2]);
    frame.r[3] = worldPosition;
}

// Scene::EvaluateIntersection, Scene.cpp:305-365
// kLean: the scene class the shading kernels are specialised for (rtgpu_upload_scene decides), two independent properties:
//   RT_LEAN(k)     -- only mesh shapes (no analytic shapes, no finite lights among the objects: a hit can only be a mesh triangle), only
//                     diffuse materials, only background / directional lights;
//   RT_TEXTURED(k) -- textures may be present (albedo / roughness / ... maps, normal maps, an environment map).
// 0 = anything, 1 = lean without textures (the Sponza-class benchmark), 2 = lean with textures (a textured Sponza), 3 = anything without
// textures (Cornell box: analytic shapes, area light, all BSDFs).
#define RT_LEAN(k) ((k) == 1 || (k) == 2 || (k) == 4)
#define RT_TEXTURED(k) ((k) == 0 || (k) == 2 || (k) == 4)
template <int kLean>
__device__ __forceinline__ static void sceneEvaluateIntersection(const RtSceneDesc& d, const Ray& ray, const Hit& hit, Intersection& out, Counters& cnt)
{
    const RtObject& obj = d.objects[hit.objectId];
    const M4 transform = loadM4(obj.transform);
    const M4 invTransform = fastInverseNoScale(transform);
    const V4 worldPosition = rayAt(ray, hit.distance);
    out.frame.r[3] = transformPoint(invTransform, worldPosition);

    if (RT_LEAN(kLean))
    {
        out.material = obj.materialIndex;
        meshEvaluateIntersection(d, d.meshes[obj.meshIndex], hit, out); cnt.c[C_MESH_HITS]++;
    }
    else if (obj.objectKind == RT_OBJECT_LIGHT)       // LightSceneObject::EvaluateIntersection, SceneObject_Light.cpp:62-73
    {
        const RtLight& L = d.lights[obj.lightIndex];
        shapeEvaluateIntersection(L.shapeKind, L.shapeParam, L.shapeParam2, out);
        cnt.c[C_ANALYTIC_HITS]++;
    }
    else
    {
        out.material = obj.materialIndex;        // ShapeSceneObject::EvaluateIntersection, SceneObject_Shape.cpp:60-64
        if (obj.shapeKind == RT_SHAPE_MESH) { meshEvaluateIntersection(d, d.meshes[obj.meshIndex], hit, out); cnt.c[C_MESH_HITS]++; }
        else { shapeEvaluateIntersection(obj.shapeKind, obj.shapeParam, obj.shapeParam2, out); cnt.c[C_ANALYTIC_HITS]++; }
    }

    const bool mapped = RT_TEXTURED(kLean) && out.material != RT_NO_MATERIAL && d.materials[out.material].normalMapTexture != RT_NO_TEXTURE;   // normal mapping, :327-337
    V4 localNormal = zero4();
    if (mapped) localNormal = materialGetNormalVector<kLean>(d, d.materials[out.material], out.texCoord);
    composeShadingFrame(transform, worldPosition, out.frame.r[0], out.frame.r[2], mapped, localNormal, out.frame);
}

// =====================================================================================================
// Lights -- Core/Scene/Light/*.cpp
// =====================================================================================================
struct IlluminateResult { V4 directionToLight; float distance, directPdfW, cosAtLight; };

// ILight::Illuminate; returns radiance (4 lanes)
// kLean: the scene's lights are background and directional lights only
// BackgroundLight::GetBackgroundColor, BackgroundLight.cpp:45-61 (!RT_TEXTURED: no textures in the scene)
template <int kLean>
RT_DEV V4 backgroundColor(const RtSceneDesc& d, const RtLight& L, V4 dir)
{
    V4 color = load4(L.color);
    if (RT_TEXTURED(kLean) && L.texture != RT_NO_TEXTURE) color = color * max4(zero4(), textureEvaluateK<kLean>(d, L.texture, cartesianToSpherical(dir)));
    return color;
}
template <int kLean>
__device__ __forceinline__ static V4 lightIlluminate(const RtSceneDesc& d, const RtLight& L, const Intersection& isect, const float u[3], IlluminateResult& out)
{
    const uint32_t lightType = RT_LEAN(kLean) ? (L.type == RT_LIGHT_BACKGROUND ? (uint32_t)RT_LIGHT_BACKGROUND : (uint32_t)RT_LIGHT_DIRECTIONAL) : L.type;
    out.directionToLight = zero4(); out.distance = -1.0f; out.directPdfW = -1.0f; out.cosAtLight = -1.0f;   // Light.h:64-71
    const V4 color = load4(L.color);
    switch (lightType)
    {
    case RT_LIGHT_AREA:          // AreaLight.cpp:55-107 (rendererSupportsSolidAngleSampling = true)
    {
        const M4 worldToLight = loadM4(L.invTransform), lightToWorld = loadM4(L.transform);
        const V4 ref = transformPoint(worldToLight, isect.frame.r[3]);
        ShapeSample s;
        if (!shapeSampleFrom(L.shapeKind, L.shapeParam, ref, u, s)) return zero4();
        out.directionToLight = transformVector(lightToWorld, s.direction);
        out.distance = s.distance; out.cosAtLight = s.cosAtSurface; out.directPdfW = s.pdf;
        return color;
    }
    case RT_LIGHT_BACKGROUND:    // BackgroundLight.cpp:63-76
    {
        const V4 dirLocal = getHemisphere(u[0], u[1]);
        out.directionToLight = localToWorld(isect, dirLocal);
        out.directPdfW = uniformHemispherePdf();
        out.distance = FLT_MAX;
        out.cosAtLight = 1.0f;
        return backgroundColor<kLean>(d, L, out.directionToLight);
    }
    case RT_LIGHT_DIRECTIONAL:   // DirectionalLight.cpp:48-92
    {
        V4 dir = zero4();
        if (L.isDelta) { out.directPdfW = 1.0f; dir = V4(0, 0, 1, 0); }
        else
        {
            out.directPdfW = sphereCapPdf(L.cosAngle);
            const float phi = RTD_2PI * u[1];
            const V4 sinCosPhi = sinCos(phi);
            float cosTheta = Lerp(L.cosAngle, 1.0f, u[0]);
            float sinThetaSqr = 1.0f - Sqr(cosTheta);
            float sinTheta = sqrtf(sinThetaSqr);
            dir.x = sinTheta * sinCosPhi.x; dir.y = sinTheta * sinCosPhi.y; dir.z = cosTheta;
            dir = normalized3(dir);
        }
        out.directionToLight = transformVectorNeg(loadM4(L.transform), dir);
        out.cosAtLight = 1.0f;
        out.distance = FLT_MAX;
        return color;
    }
    case RT_LIGHT_POINT:         // PointLight.cpp:35-49
    {
        out.directionToLight = load4(L.transform + 12) - isect.frame.r[3];
        const float sqrDistance = sqrLength3(out.directionToLight);
        out.directPdfW = sqrDistance;
        out.distance = sqrtf(sqrDistance);
        out.directionToLight = out.directionToLight / out.distance;
        out.cosAtLight = 1.0f;
        return color;
    }
    default:                     // RT_LIGHT_SPOT, SpotLight.cpp:41-61
    {
        out.directionToLight = load4(L.transform + 12) - isect.frame.r[3];
        const float sqrDistance = sqrLength3(out.directionToLight);
        out.directPdfW = sqrDistance;
        out.distance = sqrtf(sqrDistance);
        out.directionToLight = out.directionToLight / out.distance;
        out.cosAtLight = 1.0f;
        const float angle = dot3(out.directionToLight, neg(V4(0, 0, 1, 0)));
        if (angle < L.cosAngle) return zero4();
        return color;
    }
    }
}

// ILight::GetRadiance for a ray that hit / escaped; ray and hitPoint are in light space.
template <int kLean>
__device__ __forceinline__ static V4 lightGetRadiance(const RtSceneDesc& d, const RtLight& L, const Ray& lray, V4 hitPoint, float cosAtLight, float& outDirectPdfA)
{
    const uint32_t lightType = RT_LEAN(kLean) ? (L.type == RT_LIGHT_BACKGROUND ? (uint32_t)RT_LIGHT_BACKGROUND : (uint32_t)RT_LIGHT_DIRECTIONAL) : L.type;
    switch (lightType)
    {
    case RT_LIGHT_AREA:          // AreaLight.cpp:109-147
        if (cosAtLight < RTD_EPSILON) return zero4();
        outDirectPdfA = shapePdf(L.shapeKind, L.shapeParam, lray.origin, hitPoint);
        return load4(L.color);
    case RT_LIGHT_BACKGROUND:    // BackgroundLight.cpp:78-92
        outDirectPdfA = uniformHemispherePdf();
        return backgroundColor<kLean>(d, L, lray.dir);
    case RT_LIGHT_DIRECTIONAL:   // DirectionalLight.cpp:94-121
        if (L.isDelta) return zero4();
        if (dot3(lray.dir, V4(0, 0, 1, 0)) > -L.cosAngle) return zero4();
        outDirectPdfA = sphereCapPdf(L.cosAngle);
        return load4(L.color);
    default:                     // point/spot cannot be hit (ILight::GetRadiance Light.cpp:28-32 is fatal)
        return zero4();
    }
}

// =====================================================================================================
// Materials / BSDFs -- Core/Material/Material.cpp, Core/Material/BSDF/*.cpp
// =====================================================================================================
#define kCosEpsilon (1.0e-5f) // = 1.0e-5f;                       // BSDF.h:53
#define kSpecularEventRoughnessTreshold (0.005f) // = 0.005f;     // BSDF.h:57
enum { EV_NULL = 0, EV_DIFFUSE_REFLECTION = 1, EV_GLOSSY_REFLECTION = 4, EV_GLOSSY_REFRACTION = 8, EV_SPECULAR_REFLECTION = 16, EV_SPECULAR_REFRACTION = 32,
       EV_SPECULAR = 48 };   // BSDF.h:25-43

struct MatParams { V4 baseColor, emission; float roughness, metalness, IoR; };   // SampledMaterialParameters ShadingData.h:12-19

// GGX microfacet, Core/Material/BSDF/Microfacet.h:10-60 (alpha = roughness^2)
struct Microfacet
{
    float alphaSqr;
    __device__ __forceinline__ explicit Microfacet(float alpha) : alphaSqr(alpha * alpha) {}
    __device__ __forceinline__ float D(V4 m) const
    {
        const float NdotH = m.z;
        const float cosThetaSq = Sqr(NdotH);
        const float tanThetaSq = Max(1.0f - cosThetaSq, 0.0f) / cosThetaSq;
        const float cosThetaQu = cosThetaSq * cosThetaSq;
        return alphaSqr * RTD_INV_PI / (cosThetaQu * Sqr(alphaSqr + tanThetaSq));
    }
    __device__ __forceinline__ float Pdf(V4 m) const { return D(m) * Abs(m.z); }
    __device__ __forceinline__ float G(float NdotV, float NdotL) const
    {
        float tanThetaSqV = (1.0f - NdotV * NdotV) / (NdotV * NdotV);
        float tanThetaSqL = (1.0f - NdotL * NdotL) / (NdotL * NdotL);
        return 4.0f / ((1.0f + sqrtf(1.0f + alphaSqr * tanThetaSqV)) * (1.0f + sqrtf(1.0f + alphaSqr * tanThetaSqL)));
    }
    __device__ __forceinline__ V4 Sample(float ux, float uy) const
    {
        const float cosThetaSqr = (1.0f - ux) / (1.0f + (alphaSqr - 1.0f) * ux);
        const float cosTheta = sqrtf(cosThetaSqr);
        const float sinTheta = sqrtf(1.0f - cosThetaSqr);
        const float phi = RTD_2PI * uy;
        const V4 xy = sinTheta * sinCos(phi);
        return V4(xy.x, xy.y, cosTheta, xy.w);   // Select<0,0,1,0>(xy, splat(cosTheta))
    }
};

struct BsdfSample { V4 color, incomingDir; float pdf; uint32_t event; };

RT_DEV bool bsdfSampleImpl(uint32_t bsdf, const RtMaterial& mat, const MatParams& mp, const float u[3], V4 outgoingDir, BsdfSample& out);

RT_DEV bool bsdfSampleMetal(const RtMaterial& mat, const MatParams& mp, V4 outgoingDir, BsdfSample& out)   // MetalBSDF.cpp:15-35
{
    const float NdotV = outgoingDir.z;
    if (NdotV < kCosEpsilon) return false;
    const float F = fresnelMetal(NdotV, mat.IoR, mat.K);
    out.color = mp.baseColor * splat(F);
    out.incomingDir = neg(reflect3(outgoingDir, V4(0, 0, 1, 0)));
    out.pdf = 1.0f;
    out.event = EV_SPECULAR_REFLECTION;
    return true;
}
RT_DEV bool bsdfSampleDielectric(const MatParams& mp, const float u[3], V4 outgoingDir, BsdfSample& out)   // DielectricBSDF.cpp:15-103
{
    const float NdotV = outgoingDir.z;
    if (Abs(NdotV) < kCosEpsilon) return false;
    const float ior = mp.IoR;
    const float F = fresnelDielectric(NdotV, ior);
    const float minReflectionProbability = 0.25f;
    const float reflectionProbability = minReflectionProbability + (1.0f - minReflectionProbability) * F;
    const float refractionProbability = 1.0f - reflectionProbability;
    const bool reflection = (reflectionProbability >= 1.0f) || u[0] < reflectionProbability;
    if (reflection) { out.incomingDir = neg(reflect3(outgoingDir, V4(0, 0, 1, 0))); out.event = EV_SPECULAR_REFLECTION; }
    else { out.incomingDir = refract3(neg(outgoingDir), V4(0, 0, 1, 0), ior); out.event = EV_SPECULAR_REFRACTION; }
    const float NdotL = out.incomingDir.z;
    if ((NdotV * NdotL > 0.0f) != reflection) return false;
    if (reflection) { out.pdf = reflectionProbability; out.color = splat(1.0f); out.color = out.color * (F / reflectionProbability); }
    else { out.pdf = refractionProbability; out.color = mp.baseColor; out.color = out.color * ((1.0f - F) / refractionProbability); }
    return true;
}
RT_DEV bool bsdfSamplePlastic(const MatParams& mp, const float u[3], V4 outgoingDir, BsdfSample& out)   // PlasticBSDF.cpp:15-64
{
    const float NdotV = outgoingDir.z;
    if (NdotV < kCosEpsilon) return false;
    const float ior = mp.IoR;
    const float Fi = fresnelDielectric(NdotV, ior);
    const float minSpecularWeight = 0.25f;
    const float specularWeight = minSpecularWeight + Fi * (1.0f - minSpecularWeight);
    const float diffuseWeight = (1.0f - Fi) * colorMax(mp.baseColor);
    const float specularProbability = specularWeight / (specularWeight + diffuseWeight);
    const float diffuseProbability = 1.0f - specularProbability;
    const bool specular = (specularProbability >= 1.0f) || (u[2] < specularProbability);
    if (specular)
    {
        out.color = splat(Fi / specularProbability);
        out.incomingDir = neg(reflect3(outgoingDir, V4(0, 0, 1, 0)));
        out.pdf = specularProbability;
        out.event = EV_SPECULAR_REFLECTION;
    }
    else
    {
        out.incomingDir = getHemisphereCos(u[0], u[1]);
        const float NdotL = out.incomingDir.z;
        out.pdf = out.incomingDir.z * RTD_INV_PI * diffuseProbability;
        const float Fo = fresnelDielectric(NdotL, ior);
        out.color = mp.baseColor * ((1.0f - Fi) * (1.0f - Fo) / diffuseProbability);
        out.event = EV_DIFFUSE_REFLECTION;
    }
    return true;
}

RT_DEV bool bsdfSampleImpl(uint32_t bsdf, const RtMaterial& mat, const MatParams& mp, const float u[3], V4 outgoingDir, BsdfSample& out)
{
    out.color = zero4(); out.incomingDir = zero4(); out.pdf = 0.0f; out.event = EV_NULL;   // BSDF.h:71-75
    switch (bsdf)
    {
    case RT_BSDF_NULL: return false;                              // NullBSDF.cpp:11-16
    case RT_BSDF_DIFFUSE:                                         // DiffuseBSDF.cpp:14-29
    {
        const float NdotV = outgoingDir.z;
        if (NdotV < kCosEpsilon) return false;
        out.incomingDir = getHemisphereCos(u[0], u[1]);
        out.pdf = out.incomingDir.z * RTD_INV_PI;
        out.color = mp.baseColor;
        out.event = EV_DIFFUSE_REFLECTION;
        return true;
    }
    case RT_BSDF_ROUGH_DIFFUSE:                                   // RoughDiffuseBSDF.cpp:14-49
    {
        const float NdotV = outgoingDir.z;
        if (NdotV < kCosEpsilon) return false;
        out.incomingDir = getHemisphereCos(u[0], u[1]);
        const float NdotL = out.incomingDir.z;
        const float LdotV = Max(0.0f, dot3(outgoingDir, neg(out.incomingDir)));
        const float roughness = mp.roughness;
        const float s2 = roughness * roughness;
        const float A = 1.0f - 0.50f * s2 / (0.33f + s2);
        const float B = 0.45f * s2 / (0.09f + s2);
        const float s = LdotV - NdotL * NdotV;
        const float stinv = s > 0.0f ? s / Max(NdotL, NdotV) : 0.0f;
        const float value = Max(A + B * stinv, 0.0f);
        out.pdf = NdotL * RTD_INV_PI;
        out.color = mp.baseColor * value;
        out.event = EV_DIFFUSE_REFLECTION;
        return true;
    }
    case RT_BSDF_DIELECTRIC: return bsdfSampleDielectric(mp, u, outgoingDir, out);
    case RT_BSDF_ROUGH_DIELECTRIC:                                // RoughDielectricBSDF.cpp:17-115
    {
        const float NdotV = outgoingDir.z;
        if (Abs(NdotV) < kCosEpsilon) return false;
        const float ior = mp.IoR;
        const float roughness = mp.roughness;
        if (roughness < kSpecularEventRoughnessTreshold) return bsdfSampleDielectric(mp, u, outgoingDir, out);
        const Microfacet microfacet(roughness * roughness);
        const V4 m = microfacet.Sample(u[0], u[1]);
        const float microfacetPdf = microfacet.Pdf(m);
        const float VdotH = dot3(m, outgoingDir);
        const float F = fresnelDielectric(VdotH, ior);
        const bool reflection = u[2] < F;
        if (reflection) { out.incomingDir = neg(reflect3(outgoingDir, m)); out.event = EV_GLOSSY_REFLECTION; }
        else { out.incomingDir = refract3(neg(outgoingDir), m, ior); out.event = EV_GLOSSY_REFRACTION; }
        const float NdotL = out.incomingDir.z;
        const float LdotH = dot3(m, out.incomingDir);
        if ((NdotV * NdotL > 0.0f) != reflection) return false;
        const float D = microfacet.D(m);
        const float G = microfacet.G(NdotV, NdotL);
        out.color = splat(Abs(VdotH) * G * D / (microfacetPdf * Abs(NdotV)));
        if (reflection) out.pdf = F * microfacetPdf / (4.0f * Abs(VdotH));
        else
        {
            const float eta = NdotV < 0.0f ? ior : 1.0f / ior;
            const float denom = Sqr(eta * VdotH + LdotH);
            out.pdf = (1.0f - F) * microfacetPdf * Abs(LdotH) / denom;
            out.color = out.color * mp.baseColor;
        }
        return true;
    }
    case RT_BSDF_METAL: return bsdfSampleMetal(mat, mp, outgoingDir, out);
    case RT_BSDF_ROUGH_METAL:                                     // RoughMetalBSDF.cpp:17-65
    {
        const float roughness = mp.roughness;
        if (roughness < kSpecularEventRoughnessTreshold) return bsdfSampleMetal(mat, mp, outgoingDir, out);
        const float NdotV = outgoingDir.z;
        if (NdotV < kCosEpsilon) return false;
        const Microfacet microfacet(roughness * roughness);
        const V4 m = microfacet.Sample(u[0], u[1]);
        out.incomingDir = neg(reflect3(outgoingDir, m));
        if (out.incomingDir.z < kCosEpsilon) return false;
        const float NdotL = out.incomingDir.z;
        const float VdotH = dot3(m, outgoingDir);
        const float pdf = microfacet.Pdf(m);
        const float D = microfacet.D(m);
        const float G = microfacet.G(NdotV, NdotL);
        const float F = fresnelMetal(VdotH, mat.IoR, mat.K);
        out.pdf = pdf / (4.0f * VdotH);
        out.color = mp.baseColor * splat(VdotH * F * G * D / (pdf * NdotV));
        out.event = EV_GLOSSY_REFLECTION;
        return true;
    }
    case RT_BSDF_PLASTIC: return bsdfSamplePlastic(mp, u, outgoingDir, out);
    default:                                                      // RT_BSDF_ROUGH_PLASTIC, RoughPlasticBSDF.cpp:18-88
    {
        const float NdotV = outgoingDir.z;
        if (NdotV < kCosEpsilon) return false;
        const float roughness = mp.roughness;
        if (roughness < kSpecularEventRoughnessTreshold) return bsdfSamplePlastic(mp, u, outgoingDir, out);
        const float ior = mp.IoR;
        const float Fi = fresnelDielectric(NdotV, ior);
        const float specularWeight = Fi;
        const float diffuseWeight = (1.0f - Fi) * colorMax(mp.baseColor);
        const float specularProbability = specularWeight / (specularWeight + diffuseWeight);
        const float diffuseProbability = 1.0f - specularProbability;
        const bool specular = u[2] < specularProbability;
        if (specular)
        {
            const Microfacet microfacet(roughness * roughness);
            const V4 m = microfacet.Sample(u[0], u[1]);
            out.incomingDir = neg(reflect3(outgoingDir, m));
            const float NdotL = out.incomingDir.z;
            const float VdotH = dot3(m, outgoingDir);
            if (NdotL < kCosEpsilon || VdotH < kCosEpsilon) return false;
            const float pdf = microfacet.Pdf(m);
            const float D = microfacet.D(m);
            const float G = microfacet.G(NdotV, NdotL);
            const float F = fresnelDielectric(VdotH, mat.IoR);
            out.pdf = pdf / (4.0f * VdotH) * specularProbability;
            out.color = splat(VdotH * F * G * D / (pdf * NdotV * specularProbability));
            out.event = EV_GLOSSY_REFLECTION;
        }
        else
        {
            out.incomingDir = getHemisphereCos(u[0], u[1]);
            const float NdotL = out.incomingDir.z;
            out.pdf = out.incomingDir.z * RTD_INV_PI * diffuseProbability;
            const float Fo = fresnelDielectric(NdotL, ior);
            out.color = mp.baseColor * ((1.0f - Fi) * (1.0f - Fo) / diffuseProbability);
            out.event = EV_DIFFUSE_REFLECTION;
        }
        return true;
    }
    }
}

RT_DEV V4 bsdfEvaluatePlastic(const MatParams& mp, V4 outgoingDir, V4 incomingDir, float& outPdf, float* outRev = nullptr)   // PlasticBSDF.cpp:66-99
{
    const float NdotV = outgoingDir.z;
    const float NdotL = -incomingDir.z;
    if (NdotV < kCosEpsilon || NdotL < kCosEpsilon) return zero4();
    const float ior = mp.IoR;
    const float Fi = fresnelDielectric(NdotV, ior);
    const float Fo = fresnelDielectric(NdotL, ior);
    const float specularWeight = Fi;
    const float diffuseWeight = (1.0f - Fi) * colorMax(mp.baseColor);
    const float specularProbability = specularWeight / (specularWeight + diffuseWeight);
    const float diffuseProbability = 1.0f - specularProbability;
    outPdf = NdotL * RTD_INV_PI * diffuseProbability;
    if (outRev) *outRev = NdotV * RTD_INV_PI * diffuseProbability;
    return mp.baseColor * (NdotL * RTD_INV_PI * (1.0f - Fi) * (1.0f - Fo));
}

// BSDF::Evaluate.  outPdf is left untouched on the early-out paths exactly like the reference (the
// caller only reads it when the returned colour is not AlmostZero).
// outRev = *outReversePdfW (bidirectional integrator only).  The reference leaves it unwritten when RoughPlasticBSDF falls back to
// PlasticBSDF (RoughPlasticBSDF.cpp:95-98, an uninitialised read in its callers); here that case gets PlasticBSDF's reverse pdf.
RT_DEV V4 bsdfEvaluate(uint32_t bsdf, const RtMaterial& mat, const MatParams& mp, V4 outgoingDir, V4 incomingDir, float& outPdf, float* outRev = nullptr)
{
    switch (bsdf)
    {
    case RT_BSDF_NULL: return zero4();
    case RT_BSDF_DIFFUSE:                                         // DiffuseBSDF.cpp:31-54
    {
        const float NdotV = outgoingDir.z, NdotL = -incomingDir.z;
        if (NdotV > kCosEpsilon && NdotL > kCosEpsilon)
        {
            outPdf = NdotL * RTD_INV_PI;
            if (outRev) *outRev = NdotV * RTD_INV_PI;
            return mp.baseColor * splat(NdotL * RTD_INV_PI);
        }
        return zero4();
    }
    case RT_BSDF_ROUGH_DIFFUSE:                                   // RoughDiffuseBSDF.cpp:51-73
    {
        const float NdotV = outgoingDir.z, NdotL = -incomingDir.z;
        if (NdotV > kCosEpsilon && NdotL > kCosEpsilon)
        {
            outPdf = NdotL * RTD_INV_PI;
            if (outRev) *outRev = NdotV * RTD_INV_PI;
            const float LdotV = Max(0.0f, dot3(outgoingDir, neg(incomingDir)));
            const float roughness = mp.roughness;
            const float s2 = roughness * roughness;
            const float A = 1.0f - 0.50f * s2 / (0.33f + s2);
            const float B = 0.45f * s2 / (0.09f + s2);
            const float s = LdotV - NdotL * NdotV;
            const float stinv = s > 0.0f ? s / Max(NdotL, NdotV) : 0.0f;
            const float value = NdotL * RTD_INV_PI * Max(A + B * stinv, 0.0f);
            return mp.baseColor * value;
        }
        return zero4();
    }
    case RT_BSDF_DIELECTRIC: outPdf = 0.0f; if (outRev) *outRev = 0.0f; return zero4();       // DielectricBSDF.cpp:105-121
    case RT_BSDF_ROUGH_DIELECTRIC:                                // RoughDielectricBSDF.cpp:117-193
    {
        const float NdotV = outgoingDir.z, NdotL = -incomingDir.z;
        if (Abs(NdotV) < kCosEpsilon || Abs(NdotL) < kCosEpsilon) return zero4();
        const float roughness = mp.roughness;
        if (roughness < kSpecularEventRoughnessTreshold) return zero4();
        const float ior = mp.IoR;
        const float eta = NdotV < 0.0f ? ior : 1.0f / ior;
        const bool reflection = NdotV * NdotL >= 0.0f;
        V4 m;
        if (reflection) m = outgoingDir - incomingDir; else m = eta * outgoingDir - incomingDir;
        m = m * Signum(m.z);
        m = normalized3(m);
        if (Abs(m.z) < kCosEpsilon) return zero4();
        const float VdotH = dot3(m, outgoingDir);
        const float LdotH = dot3(m, neg(incomingDir));
        float color, pdf;
        const Microfacet microfacet(roughness * roughness);
        const float F = fresnelDielectric(VdotH, ior);
        const float D = microfacet.D(m);
        const float G = microfacet.G(NdotV, NdotL);
        if (reflection)
        {
            pdf = F * microfacet.Pdf(m) / (4.0f * Abs(VdotH));
            color = F * G * D / (4.0f * Abs(NdotV));
        }
        else
        {
            const float denom = Sqr(eta * VdotH + LdotH);
            pdf = (1.0f - F) * microfacet.Pdf(m) * Abs(LdotH) / denom;
            color = Abs(VdotH * LdotH) * (1.0f - F) * G * D / (denom * Abs(NdotV));
        }
        outPdf = pdf;
        if (outRev) *outRev = pdf;
        return splat(color);
    }
    case RT_BSDF_METAL: outPdf = 0.0f; if (outRev) *outRev = 0.0f; return zero4();            // MetalBSDF.cpp:37-54
    case RT_BSDF_ROUGH_METAL:                                     // RoughMetalBSDF.cpp:67-107
    {
        const float roughness = mp.roughness;
        if (roughness < kSpecularEventRoughnessTreshold) { outPdf = 0.0f; return zero4(); }
        const V4 m = normalized3(outgoingDir - incomingDir);
        const float NdotV = outgoingDir.z, NdotL = -incomingDir.z;
        const float VdotH = dot3(m, outgoingDir);
        if (NdotV < kCosEpsilon || NdotL < kCosEpsilon || VdotH < kCosEpsilon) return zero4();
        const Microfacet microfacet(roughness * roughness);
        const float D = microfacet.D(m);
        const float G = microfacet.G(NdotV, NdotL);
        const float F = fresnelMetal(VdotH, mat.IoR, mat.K);
        outPdf = microfacet.Pdf(m) / (4.0f * VdotH);
        if (outRev) *outRev = outPdf;
        return mp.baseColor * splat(F * G * D / (4.0f * NdotV));
    }
    case RT_BSDF_PLASTIC: return bsdfEvaluatePlastic(mp, outgoingDir, incomingDir, outPdf, outRev);
    default:                                                      // RT_BSDF_ROUGH_PLASTIC, RoughPlasticBSDF.cpp:90-158
    {
        const float roughness = mp.roughness;
        if (roughness < kSpecularEventRoughnessTreshold) return bsdfEvaluatePlastic(mp, outgoingDir, incomingDir, outPdf, outRev);
        const float NdotV = outgoingDir.z, NdotL = -incomingDir.z;
        if (NdotV < kCosEpsilon || NdotL < kCosEpsilon) return zero4();
        const float ior = mp.IoR;
        const float Fi = fresnelDielectric(NdotV, ior);
        const float Fo = fresnelDielectric(NdotL, ior);
        const float specularWeight = Fi;
        const float diffuseWeight = (1.0f - Fi) * colorMax(mp.baseColor);
        const float specularProbability = specularWeight / (specularWeight + diffuseWeight);
        const float diffuseProbability = 1.0f - specularProbability;
        float diffusePdf = NdotL * RTD_INV_PI;
        float specularPdf = 0.0f;
        V4 diffuseTerm = mp.baseColor * (NdotL * RTD_INV_PI * (1.0f - Fi) * (1.0f - Fo));
        V4 specularTerm = zero4();
        {
            const V4 m = normalized3(outgoingDir - incomingDir);
            const float VdotH = dot3(m, outgoingDir);
            if (VdotH >= kCosEpsilon)
            {
                const Microfacet microfacet(roughness * roughness);
                const float D = microfacet.D(m);
                const float G = microfacet.G(NdotV, NdotL);
                const float F = fresnelDielectric(VdotH, mat.IoR);
                specularPdf = microfacet.Pdf(m) / (4.0f * VdotH);
                specularTerm = splat(F * G * D / (4.0f * NdotV));
            }
        }
        outPdf = diffusePdf * diffuseProbability + specularPdf * specularProbability;
        if (outRev) *outRev = (NdotV * RTD_INV_PI) * diffuseProbability + specularPdf * specularProbability;
        return diffuseTerm + specularTerm;
    }
    }
}

struct ShadingData { Intersection intersection; V4 outgoingDirWorldSpace; MatParams mp; };   // ShadingData.h:21-30

// Material::EvaluateShadingData, Material.cpp:151-158; MaterialParameter<T>::Evaluate, MaterialParameter.h:22-32:
// value = baseValue * texture->Evaluate(uv) (Vector4 parameters: all four lanes; float parameters: lane x).
// !RT_TEXTURED(kLean): the scene has no textures.
template <int kLean>
RT_DEV void materialEvaluateShadingData(const RtSceneDesc& d, const RtMaterial& mat, ShadingData& sd)
{
    sd.mp.baseColor = load4(mat.baseColor);
    sd.mp.emission = load4(mat.emission);
    sd.mp.roughness = mat.roughness;
    sd.mp.metalness = mat.metalness;
    sd.mp.IoR = mat.IoR;
    if (!RT_TEXTURED(kLean)) return;
    const V4 uv = sd.intersection.texCoord;
    if (mat.baseColorTexture != RT_NO_TEXTURE) sd.mp.baseColor = sd.mp.baseColor * textureEvaluateK<kLean>(d, mat.baseColorTexture, uv);
    if (mat.emissionTexture != RT_NO_TEXTURE) sd.mp.emission = sd.mp.emission * textureEvaluateK<kLean>(d, mat.emissionTexture, uv);
    if (mat.roughnessTexture != RT_NO_TEXTURE) sd.mp.roughness = (splat(mat.roughness) * textureEvaluateK<kLean>(d, mat.roughnessTexture, uv)).x;
    if (mat.metalnessTexture != RT_NO_TEXTURE) sd.mp.metalness = (splat(mat.metalness) * textureEvaluateK<kLean>(d, mat.metalnessTexture, uv)).x;
}
// Material::Evaluate, Material.cpp:160-180
// kLean: every material of the scene uses the diffuse BSDF
template <int kLean>
__device__ __forceinline__ static V4 materialEvaluate(const RtMaterial& mat, const ShadingData& sd, V4 incomingDirWorldSpace, float& outPdfW, float* outRevPdfW = nullptr)
{
    const V4 incomingLocal = worldToLocal(sd.intersection, incomingDirWorldSpace);
    const V4 outgoingLocal = worldToLocal(sd.intersection, sd.outgoingDirWorldSpace);
    return bsdfEvaluate(RT_LEAN(kLean) ? (uint32_t)RT_BSDF_DIFFUSE : mat.bsdf, mat, sd.mp, outgoingLocal, incomingLocal, outPdfW, outRevPdfW);
}
// Material::Sample, Material.cpp:182-232
template <int kLean>
__device__ __forceinline__ static V4 materialSample(const RtMaterial& mat, const ShadingData& sd, const float u[3], V4& outIncomingDirWorldSpace, float& outPdfW, uint32_t& outEvent)
{
    BsdfSample s;
    const V4 outgoingLocal = worldToLocal(sd.intersection, sd.outgoingDirWorldSpace);
    if (!bsdfSampleImpl(RT_LEAN(kLean) ? (uint32_t)RT_BSDF_DIFFUSE : mat.bsdf, mat, sd.mp, u, outgoingLocal, s)) { outEvent = EV_NULL; return zero4(); }
    outIncomingDirWorldSpace = localToWorld(sd.intersection, s.incomingDir);
    outPdfW = s.pdf;
    outEvent = s.event;
    return s.color;
}

// =====================================================================================================
// Camera -- Camera::GenerateRay, Core/Scene/Camera.cpp:81-118: pinhole, barrel distortion (:86-91), thin lens with the three
// bokeh shapes of Camera::GenerateBokeh (:195-216)
// =====================================================================================================
// Everything up to (not including) the Ray constructor: k_generate stores origin + direction in the path records and every
// kernel that needs the ray re-runs makeRay from them.
RT_DEV void cameraGenerateRayParts(const RtCamera& cam, V4 coords, Sampler& sampler, V4& origin, V4& direction)
{
    const M4 transform = loadM4(cam.localToWorld);
    V4 offsetedCoords = mulSub(coords, 2.0f, splat(1.0f));   // UnipolarToBipolar Vector4ImplSSE.h:598-601
    if (cam.barrelDistortionVariableFactor != 0.0f)   // Random::GetFloat, Random.cpp:54-59
    {
        V4 radius = splat(dot2(offsetedCoords, offsetedCoords));
        const float rnd = __uint_as_float((sampler.fallbackInt() & 0x007fffffu) | 0x3f800000u) - 1.0f;
        radius = radius * (cam.barrelDistortionConstFactor + cam.barrelDistortionVariableFactor * rnd);
        offsetedCoords = mulAdd(offsetedCoords, radius, offsetedCoords);
    }
    origin = transform.r[3];
    direction = mulAdd(mulAdd(transform.r[0], offsetedCoords.x * cam.aspectRatio, transform.r[1] * offsetedCoords.y), cam.tanHalfFoV, transform.r[2]);
    if (cam.dofEnable)
    {
        const V4 focusPoint = mulAdd(direction, cam.focalPlaneDistance, origin);
        const float sx = sampler.getFloat(); const float sy = sampler.getFloat();
        // circle, hexagon (the third sample coordinate is always 0: its first rhombus, SamplingHelpers.cpp:40-57), square
        V4 bokeh;
        if (cam.bokehShape == 1u) bokeh = V4(sx * -1.0f + sy * 0.5f, sx * 0.0f + sy * 0.8660254f, 0.0f, 0.0f);
        else if (cam.bokehShape == 2u) bokeh = mulSub(V4(sx, sy, 0.0f, 0.0f), 2.0f, splat(1.0f));
        else bokeh = getCircle(sx, sy);
        const V4 randomPointOnCircle = bokeh * cam.aperture;
        origin = mulAdd(splat(randomPointOnCircle.x), transform.r[0], origin);
        origin = mulAdd(splat(randomPointOnCircle.y), transform.r[1], origin);
        direction = focusPoint - origin;
    }
}
RT_DEV Ray cameraGenerateRay(const RtCamera& cam, V4 coords, Sampler& sampler)
{
    V4 origin, direction;
    cameraGenerateRayParts(cam, coords, sampler, origin, direction);
    return makeRay(origin, direction);
}


// =====================================================================================================
// Post-processing -- Viewport::PostProcessTile (Core/Rendering/Viewport.cpp:495-550), lane-wise over rgb(+w)
// =====================================================================================================
// FastLog(Vector4) -- the FMA form, Core/Math/Transcendental.cpp:215-233 (the scalar FastLog is NOT fused)
RT_DEV float fastLogVec(float x)
{
    int32_t xi; __builtin_memcpy(&xi, &x, 4);
    const int32_t e = (int32_t)((uint32_t)(xi - 0x3f2aaaab) & 0xff800000u);
    const int32_t mi = xi - e; float m; __builtin_memcpy(&m, &mi, 4);
    const float i = (float)e * 1.19209290e-7f;
    const float f = m - 1.0f;
    const float s = f * f;
    float r = __fmaf_rn(f, 0.230836749f, -0.279208571f);
    const float t = __fmaf_rn(f, 0.331826031f, -0.498910338f);
    r = __fmaf_rn(r, s, t);
    r = __fmaf_rn(r, s, f);
    return __fmaf_rn(i, 0.693147182f, r);
}
// FastExp(Vector4), Transcendental.cpp:147-165
RT_DEV float fastExpVec(float a)
{
    const float t = a * 1.442695041f;
    const float fi = floorf(t);
    const int32_t i = __float2int_rn(fi);
    const float f = t - fi;
    float y = __fmaf_rn(f, 0.3371894346f, 0.657636276f);
    y = __fmaf_rn(f, y, 1.00172476f);
    int32_t yi; __builtin_memcpy(&yi, &y, 4);
    yi += (int32_t)((uint32_t)i << 23);
    __builtin_memcpy(&y, &yi, 4);
    if ((0.0f - a) >= 87.0f) y = 0.0f;
    if (a >= 87.0f) y = __builtin_inff();
    return y;
}
// Convert_Linear_To_sRGB, Core/Color/ColorHelpers.h:29-43
RT_DEV float linearToSrgb(float c)
{
    const float s1 = sqrtf(c), s2 = sqrtf(s1), s3 = sqrtf(s2);
    float r = 0.585122381f * s1;
    r = __fmaf_rn(s2, 0.783140355f, r);
    r = __fmaf_rn(s3, -0.368262736f, r);
    return sseMin(1.0f, sseMax(0.0f, r));   // Saturate = Min(1, Max(0, v))
}
// Vector4::FastReciprocal (Vector4ImplSSE.h:380-386): _mm_rcp_ps refined by one Newton step; the approximate seed is
// replaced by the correctly rounded 1 / v (vendor specific otherwise), the refinement is kept
RT_DEV float fastReciprocal(float v)
{
    const float rcp = 1.0f / v;
    return __fmaf_rn(-(rcp * rcp), v, rcp + rcp);
}
// ToneMap, ColorHelpers.h:85-132
RT_DEV float toneMap(float c, uint32_t tonemapper)
{
    switch (tonemapper)
    {
    case RT_TONEMAPPER_CLAMPED: return linearToSrgb(c);
    case RT_TONEMAPPER_REINHARD: return linearToSrgb(c / (1.0f + c));
    case RT_TONEMAPPER_HEJL_BURGESS_DAWSON:
    {
        const float t0 = c * __fmaf_rn(c, 6.2f, 0.5f);
        const float t1 = __fmaf_rn(c, 6.2f, 1.7f);
        const float t2 = __fmaf_rn(c, t1, 0.06f);
        return t0 * fastReciprocal(t2);
    }
    default:   // ACES
    {
        const float t0 = c * __fmaf_rn(c, 2.51f, 0.03f);
        const float t1 = __fmaf_rn(c, 2.43f, 0.59f);
        const float t2 = __fmaf_rn(c, t1, 0.14f);
        return linearToSrgb(t0 * fastReciprocal(t2));
    }
    }
}
// per-pixel dithering noise in [-1, 1): a hash of (x, y, seed) -> mantissa trick of Random::GetVector4Bipolar (Random.cpp:128-139)
RT_DEV float ditherNoise(uint32_t x, uint32_t y, uint32_t channel, uint32_t seed)
{
    uint32_t h = x * 0x9E3779B1u ^ (y * 0x85EBCA77u + 0x7F4A7C15u) ^ (seed * 0xC2B2AE3Du + channel * 0x27D4EB2Fu);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    const uint32_t bits = (h & 0x007fffffu) | 0x40000000u;   // [2, 4)
    float f; __builtin_memcpy(&f, &bits, 4);
    return f - 3.0f;
}
// one pixel: r, g, b = sum-buffer value; returns 0x00RRGGBB (Vector4::ToBGR, Vector4ImplSSE.h:89-110)
RT_DEV uint32_t postProcessPixel(float r, float g, float b, uint32_t x, uint32_t y, const RtPostprocessParams& p, const float colorScale[3])
{
    const float pixelScaling = 1.0f / (float)p.numPasses;
    float c[3] = { r * pixelScaling, g * pixelScaling, b * pixelScaling };
    // saturation: Max(0, Lerp(grayscale, rgb, saturation)), Lerp = MulAndAdd(v2 - v1, w, v1)
    const float grayscale = (c[0] * 0.2126f + c[1] * 0.7152f) + (c[2] * 0.0722f + 0.0f);
    uint32_t out = 0;
    for (int k = 0; k < 3; ++k)
    {
        float v = sseMax(0.0f, __fmaf_rn(c[k] - grayscale, p.saturation, grayscale));
        v = fastExpVec(fastLogVec(v) * p.contrast);                       // contrast
        v = v * colorScale[k];                                            // exposure: colorScale = colorFilter * powf(2, exposure), computed on the host (Viewport.cpp:453)
        v = toneMap(v, p.tonemapper);
        if (p.ditheringStrength != 0.0f) v = __fmaf_rn(ditherNoise(x, y, (uint32_t)k, p.ditherSeed), p.ditheringStrength, v);
        const float scaled = v * 255.0f;
        int32_t q = (scaled != scaled) ? (int32_t)0x80000000 : (int32_t)scaled;   // cvttps2dq (NaN -> INT_MIN); values here are far from overflow
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        out |= (uint32_t)q << (16 - 8 * k);                              // r << 16 | g << 8 | b
    }
    return out;
}

} // namespace rtd
