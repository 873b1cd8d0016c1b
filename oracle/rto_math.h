// oracle/rto_math.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the slice of rt::math the PathTracerMIS hot path uses.  Every function cites the
// reference file:line it follows (paths relative to /root/reference/).  Scalar C++, no intrinsics:
// the reference's SSE/FMA semantics are spelled out lane by lane --
//   * explicit fused ops only where the reference calls MulAndAdd/MulAndSub/NegMulAndAdd/NegMulAndSub
//     (Core/Math/Vector4ImplSSE.h:329-360); this file must be compiled with -ffp-contract=off so that
//     nothing else is contracted;
//   * _mm_min_ps/_mm_max_ps operand order and NaN behaviour (Vector4ImplSSE.h:388-396);
//   * dpps summation order (x*x' + y*y') + (z*z' + 0)   (Vector4ImplSSE.h:446-459);
//   * the approximate instructions _mm_rcp_ss (FastDivide, Core/Math/Math.h:120-127) and _mm_rsqrt_ps
//     (FastNormalize3, Vector4ImplSSE.h:519-524) are replaced by correctly rounded a/b and
//     v * (1/sqrt(d)): their x86 results are vendor specific, <= 2^-11 relative.  The only
//     intrinsics in this file are those two, behind rto_set_x86_approximations (off by default): with them the
//     oracle reproduces the reference's frames and path vertices BIT FOR BIT (tests/test_reference_images.py,
//     tests/test_reference_paths.py), which pins every other line of the restatement.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
#pragma once

#include <stdint.h>
#include <math.h>
#include <float.h>
#include <string.h>
#if defined(__x86_64__) || defined(_M_X64)
#include <immintrin.h>   // the x86 approximation mode below (FastDivide, fastNormalized3)
#define RTO_HAVE_X86_APPROX 1
#endif

namespace rto {

#define RTO_EPSILON (0.000001f)      // RT_EPSILON   Core/Math/Math.h:13
#define RTO_PI (3.14159265359f)      // RT_PI        :14
#define RTO_INV_PI (0.31830988618f)  // RT_INV_PI    :16
#define RTO_2PI (6.28318530718f)     // RT_2PI       :17

struct V4
{
    float x, y, z, w;
    V4() = default;
    V4(float x_, float y_, float z_ = 0.0f, float w_ = 0.0f) : x(x_), y(y_), z(z_), w(w_) {}
    float operator[](int i) const { return (&x)[i]; }
    float& operator[](int i) { return (&x)[i]; }
};

static inline V4 splat(float s) { return V4(s, s, s, s); }
static inline V4 zero4() { return V4(0.0f, 0.0f, 0.0f, 0.0f); }
static inline V4 load3(const float* p) { return V4(p[0], p[1], p[2], 0.0f); }   // Vector4(const Float3&) Vector4ImplSSE.h:64-71
static inline V4 load4(const float* p) { return V4(p[0], p[1], p[2], p[3]); }

// lane-wise arithmetic, Vector4ImplSSE.h:246-322
static inline V4 operator+(V4 a, V4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline V4 operator-(V4 a, V4 b) { return V4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline V4 operator*(V4 a, V4 b) { return V4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline V4 operator/(V4 a, V4 b) { return V4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
static inline V4 operator*(V4 a, float b) { return V4(a.x * b, a.y * b, a.z * b, a.w * b); }
static inline V4 operator*(float a, V4 b) { return V4(b.x * a, b.y * a, b.z * a, b.w * a); }
static inline V4 operator/(V4 a, float b) { return V4(a.x / b, a.y / b, a.z / b, a.w / b); }
// unary minus is "0 - v" (Vector4ImplSSE.h:241-244): -(+0) = +0, not -0
static inline V4 neg(V4 a) { return V4(0.0f - a.x, 0.0f - a.y, 0.0f - a.z, 0.0f - a.w); }

// fused forms, Vector4ImplSSE.h:329-363
static inline V4 mulAdd(V4 a, V4 b, V4 c) { return V4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)); }
static inline V4 mulSub(V4 a, V4 b, V4 c) { return V4(fmaf(a.x, b.x, -c.x), fmaf(a.y, b.y, -c.y), fmaf(a.z, b.z, -c.z), fmaf(a.w, b.w, -c.w)); }
static inline V4 negMulAdd(V4 a, V4 b, V4 c) { return V4(fmaf(-a.x, b.x, c.x), fmaf(-a.y, b.y, c.y), fmaf(-a.z, b.z, c.z), fmaf(-a.w, b.w, c.w)); }
static inline V4 negMulSub(V4 a, V4 b, V4 c) { return V4(fmaf(-a.x, b.x, -c.x), fmaf(-a.y, b.y, -c.y), fmaf(-a.z, b.z, -c.z), fmaf(-a.w, b.w, -c.w)); }
static inline V4 mulAdd(V4 a, float b, V4 c) { return mulAdd(a, splat(b), c); }   // Vector4Impl.h:40-58
static inline V4 mulSub(V4 a, float b, V4 c) { return mulSub(a, splat(b), c); }
static inline V4 negMulAdd(V4 a, float b, V4 c) { return negMulAdd(a, splat(b), c); }

// _mm_min_ps(a,b) = a < b ? a : b ; _mm_max_ps(a,b) = a > b ? a : b  (second operand on NaN)
static inline float sseMin(float a, float b) { return a < b ? a : b; }
static inline float sseMax(float a, float b) { return a > b ? a : b; }
static inline V4 min4(V4 a, V4 b) { return V4(sseMin(a.x, b.x), sseMin(a.y, b.y), sseMin(a.z, b.z), sseMin(a.w, b.w)); }
static inline V4 max4(V4 a, V4 b) { return V4(sseMax(a.x, b.x), sseMax(a.y, b.y), sseMax(a.z, b.z), sseMax(a.w, b.w)); }
static inline float absf(float v) { uint32_t u; memcpy(&u, &v, 4); u &= 0x7fffffffu; memcpy(&v, &u, 4); return v; }
static inline V4 abs4(V4 a) { return V4(absf(a.x), absf(a.y), absf(a.z), absf(a.w)); }     // Vector4ImplSSE.h:398-401

// scalar helpers of Core/Math/Math.h
template <typename T> static inline T Min(T a, T b) { return (a < b) ? a : b; }     // Math.h:61-65
template <typename T> static inline T Max(T a, T b) { return (a < b) ? b : a; }     // Math.h:74-78
static inline float Abs(float x) { return x < 0.0f ? -x : x; }                        // Math.h:87-98
static inline float Sqr(float x) { return x * x; }                                   // Math.h:108-112
static inline float Clamp(float x, float lo, float hi) { if (x > hi) return hi; else if (x < lo) return lo; else return x; } // Math.h:163-172
static inline float Lerp(float a, float b, float w) { return a + w * (b - a); }      // Math.h:196-200
static inline float Signum(float x) { if (x > 0.0f) return 1.0f; else if (x < 0.0f) return -1.0f; else return 0.0f; } // Math.h:147-161
// FastDivide: a * _mm_rcp_ss(b) in the reference (Math.h:120-127) -> correctly rounded divide here.
// x86 approximation mode (rto_set_x86_approximations, x86-64 hosts only): the oracle evaluates the reference's two approximate instructions
// -- _mm_rcp_ss here, _mm_rsqrt_ps in FastNormalize3 -- with the HOST'S instructions, as the reference does.  It exists to show that they are the ONLY
// difference between the oracle and the reference's frames (tests/test_reference_images.py: every pixel of every fixture then agrees on a CPU of the
// family the fixtures were rendered on); the parity oracle of the device is the exact mode (the device has no such instructions).
inline int g_rtoX86Approximations = 0;
static inline float FastDivide(float a, float b)
{
#ifdef RTO_HAVE_X86_APPROX
    if (g_rtoX86Approximations) return a * _mm_cvtss_f32(_mm_rcp_ss(_mm_load_ss(&b)));
#endif
    return a / b;
}
static inline float CopySign(float x, float y)                                        // Math.h:137-144
{
    uint32_t xi, yi; memcpy(&xi, &x, 4); memcpy(&yi, &y, 4);
    xi = (0x7fffffffu & xi) | (0x80000000u & yi); memcpy(&x, &xi, 4); return x;
}

// dpps, Vector4ImplSSE.h:446-474
static inline float dot2(V4 a, V4 b) { return (a.x * b.x + a.y * b.y) + (0.0f + 0.0f); }
static inline float dot3(V4 a, V4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + 0.0f); }
static inline float dot4(V4 a, V4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// Cross3, Vector4ImplSSE.h:476-485: r = v1.yzx*v2.zxy (rounded) then fnmadd(v1.zxy, v2.yzx, r)
static inline V4 cross3(V4 a, V4 b)
{
    const float rx = a.y * b.z, ry = a.z * b.x, rz = a.x * b.y, rw = a.w * b.w;
    return V4(fmaf(-a.z, b.y, rx), fmaf(-a.x, b.z, ry), fmaf(-a.y, b.x, rz), fmaf(-a.w, b.w, rw));
}

static inline float sqrLength3(V4 a) { return dot3(a, a); }
static inline float length3(V4 a) { return sqrtf(dot3(a, a)); }                       // Vector4ImplSSE.h:497-510
// Normalize3: v / sqrt(dot3)  (all four lanes divided), Vector4ImplSSE.h:511-517
static inline V4 normalized3(V4 a) { const float l = sqrtf(dot3(a, a)); return V4(a.x / l, a.y / l, a.z / l, a.w / l); }
// FastNormalize3: v * _mm_rsqrt_ps(dot) in the reference (:519-524) -> v * (1/sqrt(dot)) here.
static inline V4 fastNormalized3(V4 a)
{
#ifdef RTO_HAVE_X86_APPROX
    if (g_rtoX86Approximations) { const float r = _mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(dot3(a, a)))); return a * r; }
#endif
    const float r = 1.0f / sqrtf(dot3(a, a)); return a * r;
}
// Reflect3, Vector4Impl.h:119-124
static inline V4 reflect3(V4 i, V4 n) { const float d = dot3(i, n); return negMulAdd(splat(d + d), n, i); }
// Orthogonalize (Gram-Schmidt), Vector4ImplSSE.h:587-591
static inline V4 orthogonalize(V4 v, V4 ref) { return negMulAdd(splat(dot3(v, ref)), ref, v); }
// AlmostEqual over ALL FOUR lanes, Vector4Impl.h:126-129
static inline bool almostZero4(V4 v)
{
    return absf(v.x - 0.0f) < RTO_EPSILON && absf(v.y - 0.0f) < RTO_EPSILON && absf(v.z - 0.0f) < RTO_EPSILON && absf(v.w - 0.0f) < RTO_EPSILON;
}
// RayColor::Max: w masked to 0 then HorizontalMax, Core/Color/RayColor.h:110-117 + Vector4ImplSSE.h:408-414
static inline float colorMax(V4 c)
{
    const float a = sseMax(c.x, c.y);   // max(v, v.yxwz) lane 0
    const float b = sseMax(c.z, 0.0f);  // lane 2 of the same op (w masked to zero)
    return sseMax(a, b);                // max(temp, temp.zwxy) lane 0
}

// Refract3, Core/Math/Vector4.cpp:7-30
static inline V4 refract3(V4 i, V4 n, float eta)
{
    float NdotV = dot3(i, n);
    if (NdotV < 0.0f) eta = 1.0f / eta;
    const float k = 1.0f - eta * eta * (1.0f - NdotV * NdotV);
    if (k <= 0.0f) return zero4();
    V4 t = negMulAdd(splat(eta * NdotV + sqrtf(k)), n, i * eta);
    if (NdotV > 0.0f) t.z = -t.z;
    return normalized3(t);
}

// ---- 4x4 matrix, row-vector convention, Core/Math/Matrix4.h --------------------------------------
struct M4 { V4 r[4]; };
static inline M4 loadM4(const float* p) { M4 m; for (int i = 0; i < 4; ++i) m.r[i] = load4(p + 4 * i); return m; }
static inline V4 transformPoint(const M4& m, V4 a)   // Matrix4.h:110-117
{
    V4 t = mulAdd(splat(a.x), m.r[0], m.r[3]);
    t = mulAdd(splat(a.y), m.r[1], t);
    t = mulAdd(splat(a.z), m.r[2], t);
    return t;
}
static inline V4 transformVector(const M4& m, V4 a)  // Matrix4.h:133-139
{
    V4 t = splat(a.x) * m.r[0];
    t = mulAdd(splat(a.y), m.r[1], t);
    t = mulAdd(splat(a.z), m.r[2], t);
    return t;
}
static inline V4 transformVectorNeg(const M4& m, V4 a) // Matrix4.h:143-149
{
    V4 t = splat(a.x) * m.r[0];
    t = negMulSub(splat(a.y), m.r[1], t);
    t = negMulAdd(splat(a.z), m.r[2], t);
    return t;
}
// Vector4::Transpose3, Vector4ImplSSE.h:577-585.  Resulting w lanes: a.w=c.x? no -- spelled out:
//   t0 = unpacklo(a,b) = [a.x b.x a.y b.y]; t1 = unpackhi(a,b) = [a.z b.z a.w b.w]
//   a' = movelh(t0,c) = [a.x b.x c.x c.y];  b' = shuffle(t0,c,(3,1,3,2)) = [a.y b.y c.y c.w]
//   c' = shuffle(t1,c,(3,2,1,0)) = [a.z b.z c.z c.w]
static inline void transpose3(V4& a, V4& b, V4& c)
{
    const V4 a0 = a, b0 = b, c0 = c;
    a = V4(a0.x, b0.x, c0.x, c0.y);
    b = V4(a0.y, b0.y, c0.y, c0.w);
    c = V4(a0.z, b0.z, c0.z, c0.w);
}
static inline M4 fastInverseNoScale(const M4& m)     // Matrix4.h:186-193
{
    M4 r = m;
    r.r[3] = V4(0.0f, 0.0f, 0.0f, 1.0f);
    transpose3(r.r[0], r.r[1], r.r[2]);
    r.r[3] = transformVectorNeg(r, m.r[3]);
    return r;
}

// ---- Ray, Core/Math/Ray.h ---------------------------------------------------------------------------
struct Ray { V4 origin, dir, invDir, originDivDir; };
// Ray(origin, direction): dir = direction.InvNormalized(invDir)  (Ray.h:23-28, Vector4Impl.h:90-98)
static inline Ray makeRay(V4 origin, V4 direction)
{
    Ray r; r.origin = origin;
    const float len = length3(direction);
    const V4 temp(direction.x, direction.y, direction.z, len);
    const V4 invTemp(1.0f / temp.x, 1.0f / temp.y, 1.0f / temp.z, 1.0f / temp.w);
    r.invDir = splat(len) * invTemp;
    r.dir = direction * invTemp.w;
    r.originDivDir = origin * r.invDir;
    return r;
}
static inline Ray makeRayUnsafe(V4 origin, V4 direction)  // Ray::BuildUnsafe Ray.h:31-39
{
    Ray r; r.origin = origin; r.dir = direction;
    r.invDir = V4(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z, 1.0f / direction.w);
    r.originDivDir = origin * r.invDir;
    return r;
}
static inline V4 rayAt(const Ray& r, float t) { return mulAdd(r.dir, t, r.origin); }  // Ray.h:41-44
static inline Ray transformRayUnsafe(const M4& m, const Ray& ray)   // Matrix4.h:245-250
{
    return makeRayUnsafe(transformPoint(m, ray.origin), transformVector(m, ray.dir));
}

// ---- transcendental approximations, Core/Math/Transcendental.cpp ---------------------------------
static inline int32_t cvtRN(float f) { return (int32_t)lrintf(f); }   // _mm_cvtps_epi32, default rounding mode
// vector Sin, one lane (Transcendental.cpp:51-76): round-to-nearest range reduction, fused Horner
static inline float sinLane(float a)
{
    const float c0 = 9.9999970197e-01f, c1 = -1.6666577756e-01f, c2 = 8.3325579762e-03f;
    const float c3 = -1.9812576647e-04f, c4 = 2.7040521217e-06f, c5 = -2.0532988642e-08f;
    const int32_t i = cvtRN(a * (1.0f / RTO_PI));
    const float x = fmaf(-(float)i, RTO_PI, a);
    const float x2 = x * x;
    float y = fmaf(c5, x2, c4);
    y = fmaf(y, x2, c3);
    y = fmaf(y, x2, c2);
    y = fmaf(y, x2, c1);
    y = fmaf(y, x2, c0);
    y *= x;
    uint32_t u; memcpy(&u, &y, 4); u ^= ((uint32_t)i << 31); memcpy(&y, &u, 4);
    return y;
}
// SinCos(x) = Sin([x, x + PI/2, 0, 0]) & mask(1,1,0,0), TranscendentalImpl.h:22-26
static inline V4 sinCos(float x) { return V4(sinLane(x + 0.0f), sinLane(x + RTO_PI / 2.0f), 0.0f, 0.0f); }
// FastACos, Transcendental.cpp:106-120 (plain float expressions, not fused)
static inline float fastACos(float x)
{
    float negate = float(x < 0);
    x = fabsf(x);
    float ret = -0.0187293f;
    ret = ret * x + 0.0742610f;
    ret = ret * x - 0.2121144f;
    ret = ret * x + 1.5707288f;
    ret = ret * sqrtf(1.0f - x);
    ret = ret - 2.0f * negate * ret;
    return negate * 3.14159265358979f + ret;
}
// scalar FastLog, Transcendental.cpp:194-214
static inline float fastLog(float x)
{
    int32_t xi; memcpy(&xi, &x, 4);
    const int32_t e = (xi - 0x3f2aaaab) & 0xff800000;
    const int32_t mi = xi - e; float m; memcpy(&m, &mi, 4);
    const float i = 1.19209290e-7f * (float)e;
    const float f = m - 1.0f;
    const float s = f * f;
    float r = 0.230836749f * f - 0.279208571f;
    float t = 0.331826031f * f - 0.498910338f;
    r = r * s + t;
    r = r * s + f;
    r = i * 0.693147182f + r;
    return r;
}
// FastATan2, Transcendental.cpp:235-259
static inline float fastATan2(float y, float x)
{
    const float ax = Abs(x), ay = Abs(y);
    const float mx = Max(ay, ax), mn = Min(ay, ax);
    const float a = mn / mx;
    const float s = a * a, c = s * a, q = s * s;
    const float t = -0.094097948f * q - 0.33213072f;
    float r = (0.024840285f * q + 0.18681418f);
    r = r * s + t;
    r = r * c + a;
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = RTO_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

// ---- geometry helpers, Core/Math/Geometry.{h,cpp} -----------------------------------------------
static inline V4 cartesianToSpherical(V4 in)   // Geometry.cpp:8-13
{
    const float theta = fastACos(Clamp(in.y, -1.0f, 1.0f));
    const float phi = Abs(in.x) > FLT_EPSILON ? fastATan2(in.z, in.x) : 0.0f;
    return V4(phi / (2.0f * RTO_PI) + 0.5f, theta / RTO_PI, 0.0f, 0.0f);
}
static inline void buildOrthonormalBasis(V4 n, V4& u, V4& v)   // Geometry.cpp:15-32 (Duff et al.)
{
    const float sign = CopySign(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    u = V4(1.0f + sign * n.x * n.x * a, sign * n.x * n.y * a, -sign * n.x);
    v = V4(n.x * n.y * a, sign + n.y * n.y * a, -n.y);
}
static inline float sphereCapPdf(float cosTheta) { return 1.0f / (RTO_2PI * (1.0f - cosTheta)); } // Geometry.h:39-42
static inline float uniformHemispherePdf() { return RTO_INV_PI / 2.0f; }                          // Geometry.h:19-22

// Intersect_BoxRay (SSE branch), Geometry.h:57-98.  box w lanes are zero (BVH::Node::GetBox masks them).
static inline bool intersectBoxRay(const Ray& ray, V4 bmin, V4 bmax, float& outDistance)
{
    const V4 tmp1 = mulSub(bmin, ray.invDir, ray.originDivDir);
    const V4 tmp2 = mulSub(bmax, ray.invDir, ray.originDivDir);
    const V4 lmin = min4(tmp1, tmp2);
    const V4 lmax = max4(tmp1, tmp2);
    // lanes 0,1 of lx/ly/lz carry lmin, lanes 2,3 carry lmax
    const float nearD = sseMax(lmin.x, sseMax(lmin.y, lmin.z));
    const float farD = sseMin(lmax.x, sseMin(lmax.y, lmax.z));
    outDistance = nearD;
    return (farD >= nearD) && (farD >= 0.0f);
}
// Intersect_BoxRay_TwoSided, Geometry.h:100-130
static inline bool intersectBoxRayTwoSided(const Ray& ray, V4 bmin, V4 bmax, float& outNear, float& outFar)
{
    const V4 tmp1 = mulSub(bmin, ray.invDir, ray.originDivDir);
    const V4 tmp2 = mulSub(bmax, ray.invDir, ray.originDivDir);
    const V4 lmin = min4(tmp1, tmp2);
    const V4 lmax = max4(tmp1, tmp2);
    outNear = sseMax(lmin.x, sseMax(lmin.y, lmin.z));
    outFar = sseMin(lmax.x, sseMin(lmax.y, lmax.z));
    return outNear < outFar;
}
// Intersect_TriangleRay (SSE branch), Geometry.h:132-168
static inline bool intersectTriangleRay(const Ray& ray, V4 v0, V4 e1, V4 e2, float& outU, float& outV, float& outT)
{
    const V4 tvec = ray.origin - v0;
    const V4 pvec = cross3(ray.dir, e2);
    const V4 qvec = cross3(tvec, e1);
    const float det = dot3(e1, pvec);
    const float u = dot3(tvec, pvec);
    const float v = dot3(ray.dir, qvec);
    const float t = dot3(e2, qvec);
    const float uv = (u + v) / det;
    outU = u / det; outV = v / det; outT = t / det;
    // mask 0xE: v > 0, t > 0, u > 0 and NOT (u + v > 1)
    return (outV > 0.0f) && (outT > 0.0f) && (outU > 0.0f) && !(uv > 1.0f);
}

// ---- Fresnel, Core/Math/Utils.cpp ---------------------------------------------------------------------
static inline float fresnelDielectric(float NdV, float eta)   // Utils.cpp:9-29
{
    if (NdV > 0.0f) eta = 1.0f / eta;
    const float c = fabsf(NdV);
    float g = eta * eta * (1.0f - NdV * NdV);
    if (g < 1.0f)
    {
        g = sqrtf(1.0f - g);
        const float A = (g - c) / (g + c);
        const float B = (c * (g + c) - 1.0f) / (c * (g - c) + 1.0f);
        return 0.5f * A * A * (1.0f + B * B);
    }
    return 1.0f;
}
static inline float fresnelMetal(float NdV, float eta, float k)   // Utils.cpp:31-39
{
    const float NdV2 = NdV * NdV;
    const float a = eta * eta + k * k;
    const float b = a * NdV2;
    const float rs = (b - (2.0f * eta * NdV) + 1.0f) / (b + (2.0f * eta * NdV) + 1.0f);
    const float rp = (a - (2.0f * eta * NdV) + NdV2) / (a + (2.0f * eta * NdV) + NdV2);
    return (rs + rp) * 0.5f;
}

// ---- sampling helpers, Core/Math/SamplingHelpers.cpp -------------------------------------------------
static inline V4 getCircle(float ux, float uy)          // :26-35
{
    const float theta = 2.0f * RTO_PI * ux;
    const float r = sqrtf(uy);
    return r * sinCos(theta);
}
static inline V4 getSphere(float ux, float uy)          // :112-126
{
    const V4 v = mulSub(V4(ux, uy, 0.0f, 0.0f), 2.0f, splat(1.0f));
    const float t = sqrtf(1.0f - v.y * v.y);
    const float theta = RTO_PI * v.x;
    V4 result = t * sinCos(theta);
    result.z = v.y;
    return result;
}
static inline V4 getHemisphere(float ux, float uy)      // :128-133
{
    V4 p = getSphere(ux, uy); p.z = Abs(p.z); return p;
}
static inline V4 getHemisphereCos(float ux, float uy)   // :135-144
{
    const float theta = 2.0f * RTO_PI * uy;
    const float r = sqrtf(ux);
    V4 result = r * sinCos(theta);
    result.z = sqrtf(1.0f - ux);
    return result;
}
static inline V4 getFloatNormal2(float ux, float uy)    // :146-150 (Box-Muller)
{
    return sqrtf(-2.0f * fastLog(ux)) * sinCos(2.0f * RTO_PI * uy);
}

// ---- integer generators ---------------------------------------------------------------------------------
static inline uint64_t murmurFmix64(uint64_t h)   // Hash(uint64) Core/Math/Math.h:269-277
{
    h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33; return h;
}
static inline uint32_t xorShift32(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }  // GenericSampler.cpp:56-62
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
// xoroshiro128+ , Random::GetLong Core/Math/Random.cpp:33-47
struct Xoroshiro { uint64_t s[2]; };
static inline uint64_t xoroshiroNext(Xoroshiro& g)
{
    const uint64_t s0 = g.s[0]; uint64_t s1 = g.s[1];
    const uint64_t result = s0 + s1;
    s1 ^= s0;
    g.s[0] = rotl64(s0, 24) ^ s1 ^ (s1 << 16);
    g.s[1] = rotl64(s1, 37);
    return result;
}

} // namespace rto
