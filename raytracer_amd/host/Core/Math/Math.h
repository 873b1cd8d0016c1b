// Host-side math of the MI355X path-tracing core.
//
// Mirrors the slice of the reference's rt::math API (Core/Math/*.h in Witek902/Raytracer) that scene
// construction and the Viewport pass prologue need.  Plain scalar C++: nothing here runs per ray -- the
// per-ray arithmetic lives in the HIP kernels behind include/rtgpu.h.
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <float.h>
#include <string.h>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#define RAYLIB_API __attribute__((visibility("default")))
// reference: Core/Common.h:95-103 (a no-op in its FINAL configuration; here a failed check reports and aborts)
#define RT_ASSERT(expression, ...) do { if (!(expression)) { fprintf(stderr, "[rt] assertion failed: %s (%s:%d)\n", #expression, __FILE__, __LINE__); abort(); } } while (0)
#define RT_EPSILON (0.000001f)
#define RT_PI (3.14159265359f)
#define RT_INV_PI (0.31830988618f)
#define RT_2PI (6.28318530718f)

namespace rt {

using uint8 = uint8_t;
using uint16 = uint16_t;
using uint32 = uint32_t;
using uint64 = uint64_t;
using int32 = int32_t;
using int64 = int64_t;

namespace math {

// Flush-to-zero / denormals-are-zero of the CALLING THREAD's SSE unit (reference: Core/Math/Math.cpp:27-43; its Demo and Tests mains
// enable it).  The host side here only runs scene construction and the pass prologue; the device keeps IEEE denormals.
RAYLIB_API void SetFlushDenormalsToZero(bool enable = true);
RAYLIB_API bool GetFlushDenormalsToZero();

constexpr float DegToRad(const float x) { return x / 180.0f * RT_PI; }   // reference: Core/Math/Math.h:49-52
constexpr float RadToDeg(const float x) { return x / RT_PI * 180.0f; }

template <typename T> constexpr T Min(const T a, const T b) { return (a < b) ? a : b; }
template <typename T> constexpr T Max(const T a, const T b) { return (a < b) ? b : a; }
template <typename T> constexpr T Sqr(const T x) { return x * x; }
template <typename T> constexpr T Clamp(const T x, const T lo, const T hi) { return x > hi ? hi : (x < lo ? lo : x); }

struct Float2
{
    float x = 0.0f, y = 0.0f;
    Float2() = default;
    explicit Float2(float s) : x(s), y(s) {}
    Float2(float x_, float y_) : x(x_), y(y_) {}
};

struct Float3
{
    float x = 0.0f, y = 0.0f, z = 0.0f;
    Float3() = default;
    explicit Float3(float s) : x(s), y(s), z(s) {}
    Float3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

// 4-element vector (reference: Core/Math/Vector4.h).  w defaults to 0 exactly like the reference's
// (x, y, z = 0, w = 0) constructor; the single-float constructor splats.
struct Vector4
{
    float x, y, z, w;
    Vector4() : x(0), y(0), z(0), w(0) {}
    explicit Vector4(float s) : x(s), y(s), z(s), w(s) {}
    Vector4(float x_, float y_, float z_ = 0.0f, float w_ = 0.0f) : x(x_), y(y_), z(z_), w(w_) {}
    explicit Vector4(const Float3& f) : x(f.x), y(f.y), z(f.z), w(0.0f) {}
    explicit Vector4(const Float2& f) : x(f.x), y(f.y), z(0.0f), w(0.0f) {}
    static Vector4 Zero() { return Vector4(); }
    float operator[](uint32 i) const { return (&x)[i]; }
    float& operator[](uint32 i) { return (&x)[i]; }
    Vector4 operator+(const Vector4& b) const { return { x + b.x, y + b.y, z + b.z, w + b.w }; }
    Vector4 operator-(const Vector4& b) const { return { x - b.x, y - b.y, z - b.z, w - b.w }; }
    Vector4 operator*(const Vector4& b) const { return { x * b.x, y * b.y, z * b.z, w * b.w }; }
    Vector4 operator/(const Vector4& b) const { return { x / b.x, y / b.y, z / b.z, w / b.w }; }
    Vector4 operator*(float b) const { return { x * b, y * b, z * b, w * b }; }
    Vector4 operator/(float b) const { return { x / b, y / b, z / b, w / b }; }
    Vector4 operator-() const { return { 0.0f - x, 0.0f - y, 0.0f - z, 0.0f - w }; }
    Vector4& operator+=(const Vector4& b) { *this = *this + b; return *this; }
    Vector4& operator*=(const Vector4& b) { *this = *this * b; return *this; }
    Vector4& operator*=(float b) { *this = *this * b; return *this; }
    static Vector4 Min(const Vector4& a, const Vector4& b) { return { a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z, a.w < b.w ? a.w : b.w }; }
    static Vector4 Max(const Vector4& a, const Vector4& b) { return { a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z, a.w > b.w ? a.w : b.w }; }
    static float Dot3(const Vector4& a, const Vector4& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
    static Vector4 Cross3(const Vector4& a, const Vector4& b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.0f }; }
    float Length3() const { return sqrtf(Dot3(*this, *this)); }
    Vector4 Normalized3() const { const float l = Length3(); return { x / l, y / l, z / l, w / l }; }
    Float3 ToFloat3() const { return { x, y, z }; }
    Float2 ToFloat2() const { return { x, y }; }
    bool IsValid() const { return std::isfinite(x) && std::isfinite(y) && std::isfinite(z) && std::isfinite(w); }
    static bool AlmostEqual(const Vector4& a, const Vector4& b, float eps = RT_EPSILON)
    {
        return fabsf(a.x - b.x) < eps && fabsf(a.y - b.y) < eps && fabsf(a.z - b.z) < eps && fabsf(a.w - b.w) < eps;
    }
};
inline Vector4 operator*(float a, const Vector4& b) { return b * a; }

static const Vector4 VECTOR_ONE = { 1.0f, 1.0f, 1.0f, 1.0f };
static const Vector4 VECTOR_X = { 1.0f, 0.0f, 0.0f, 0.0f };
static const Vector4 VECTOR_Y = { 0.0f, 1.0f, 0.0f, 0.0f };
static const Vector4 VECTOR_Z = { 0.0f, 0.0f, 1.0f, 0.0f };
static const Vector4 VECTOR_W = { 0.0f, 0.0f, 0.0f, 1.0f };
static const Vector4 VECTOR_MAX = { FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX };

// Axis aligned box (reference: Core/Math/Box.h)
struct Box
{
    Vector4 min, max;
    Box() = default;
    Box(const Vector4& mn, const Vector4& mx) : min(mn), max(mx) {}
    Box(const Vector4& a, const Vector4& b, const Vector4& c) : min(Vector4::Min(a, Vector4::Min(b, c))), max(Vector4::Max(a, Vector4::Max(b, c))) {}
    Box(const Vector4& center, float radius) : min(center - Vector4(radius)), max(center + Vector4(radius)) {}
    Box(const Box& a, const Box& b) : min(Vector4::Min(a.min, b.min)), max(Vector4::Max(a.max, b.max)) {}
    static Box Empty() { return { VECTOR_MAX, -VECTOR_MAX }; }
    static Box Full() { return { -VECTOR_MAX, VECTOR_MAX }; }
    float SurfaceArea() const { const Vector4 s = max - min; return s.x * (s.y + s.z) + s.y * s.z; }
    float Volume() const { const Vector4 s = max - min; return s.x * s.y * s.z; }
};

// 4x4 matrix, row-vector convention (reference: Core/Math/Matrix4.h)
struct Matrix4
{
    Vector4 rows[4];
    Matrix4() {}
    Matrix4(const Vector4& r0, const Vector4& r1, const Vector4& r2, const Vector4& r3) { rows[0] = r0; rows[1] = r1; rows[2] = r2; rows[3] = r3; }
    static Matrix4 Identity() { return { VECTOR_X, VECTOR_Y, VECTOR_Z, VECTOR_W }; }
    Vector4& operator[](int i) { return rows[i]; }
    const Vector4& operator[](int i) const { return rows[i]; }
    const Vector4& GetTranslation() const { return rows[3]; }
    RAYLIB_API static Matrix4 MakeTranslation(const Vector4& pos);
    RAYLIB_API static Matrix4 MakeScaling(const Vector4& scale);
    RAYLIB_API Matrix4 operator*(const Matrix4& b) const;     // (a * b): apply a first, then b
    RAYLIB_API Matrix4 Inverse() const;
    RAYLIB_API Box TransformBox(const Box& box) const;
    Vector4 TransformPoint(const Vector4& a) const { return rows[0] * a.x + rows[1] * a.y + rows[2] * a.z + rows[3]; }
    Vector4 TransformVector(const Vector4& a) const { return rows[0] * a.x + rows[1] * a.y + rows[2] * a.z; }
    bool IsValid() const { return rows[0].IsValid() && rows[1].IsValid() && rows[2].IsValid() && rows[3].IsValid(); }
    void Store(float out[16]) const { memcpy(out, rows, 64); }
};

// Unit quaternion (reference: Core/Math/Quaternion.h); q = (x, y, z, w)
struct Quaternion
{
    Vector4 q;
    Quaternion() : q(0.0f, 0.0f, 0.0f, 1.0f) {}
    explicit Quaternion(const Vector4& v) : q(v) {}
    Quaternion(float x, float y, float z, float w) : q(x, y, z, w) {}
    static Quaternion Identity() { return Quaternion(); }
    RAYLIB_API static Quaternion FromAxisAndAngle(const Vector4& axis, float angle);
    RAYLIB_API static Quaternion RotationX(float angle);
    RAYLIB_API static Quaternion RotationY(float angle);
    RAYLIB_API static Quaternion RotationZ(float angle);
    // pitch (x), yaw (y), roll (z) in radians, same composition as the reference (Quaternion.cpp:156-198)
    RAYLIB_API static Quaternion FromEulerAngles(const Float3& angles);
    RAYLIB_API Quaternion operator*(const Quaternion& b) const;
    RAYLIB_API Quaternion Normalized() const;
    RAYLIB_API Vector4 GetAxisX() const;
    RAYLIB_API Vector4 GetAxisY() const;
    RAYLIB_API Vector4 GetAxisZ() const;
    RAYLIB_API Matrix4 ToMatrix4() const;
    bool IsValid() const { return q.IsValid(); }
};

// translation + rotation (reference: Core/Math/Transform.h)
class Transform
{
public:
    Transform() {}
    explicit Transform(const Vector4& translation, const Quaternion& rotation) : mTranslation(translation), mRotation(rotation) {}
    explicit Transform(const Vector4& translation) : mTranslation(translation) {}
    explicit Transform(const Quaternion& rotation) : mRotation(rotation) {}
    const Vector4& GetTranslation() const { return mTranslation; }
    const Quaternion& GetRotation() const { return mRotation; }
    void SetTranslation(const Vector4& t) { mTranslation = t; }
    void SetRotation(const Quaternion& r) { mRotation = r; }
    bool IsValid() const { return mTranslation.IsValid() && mRotation.IsValid(); }
    Matrix4 ToMatrix4() const
    {
        Matrix4 m = mRotation.ToMatrix4();
        m.rows[3] = Vector4(mTranslation.x, mTranslation.y, mTranslation.z, 1.0f);
        return m;
    }
private:
    Vector4 mTranslation;
    Quaternion mRotation;
};

// Pseudo-random generator (reference: Core/Math/Random.h): xoroshiro128+ for scalars, two xorshift128+
// lanes for GetVector4/GetFloat2.  Unlike the reference it can be seeded: Reset(seed) derives the state
// from a splitmix64 stream, Reset() draws from the process entropy source (rt::Entropy).
class RAYLIB_API Random
{
public:
    Random();
    void Reset();
    void Reset(uint64 seed);
    void SetState(const uint64 scalarState[2], const uint64 simd4State[4]);
    uint64 GetLong();
    uint32 GetInt();
    float GetFloat();
    double GetDouble();
    Float2 GetFloat2();
    Float3 GetFloat3();
    Vector4 GetVector4();
private:
    void GetIntVector4(uint64 out[2]);
    uint64 mSeed[2];
    uint64 mSeedSimd4[2][2];   // [which][lane]
};

// process-wide entropy: /dev/urandom, or a deterministic splitmix64 stream when seeded (RT_SEED env or SetGlobalSeed)
class RAYLIB_API Entropy
{
public:
    Entropy();
    ~Entropy();
    uint32 GetInt();
    static void SetGlobalSeed(uint64 seed);   // 0 => back to /dev/urandom
private:
    int mFd;
};

// helpers of the Viewport pass prologue (reference: Core/Math/SamplingHelpers.cpp:146-150,
// Core/Math/Transcendental.cpp:51-76,194-214): Box-Muller with the reference's polynomial log / sin
RAYLIB_API Vector4 GetFloatNormal2(const Float2 u);
RAYLIB_API float FastLog(float x);
RAYLIB_API Vector4 SinCos(float x);
RAYLIB_API float Sin(float x);
RAYLIB_API float Cos(float x);

} // namespace math
} // namespace rt
