// Host-side math implementation (see Core/Math/Math.h).  Compiled with -ffp-contract=off.
#include "../Core/Math/Math.h"

#include <fcntl.h>
#include <unistd.h>
#include <stdlib.h>
#include <mutex>

namespace rt {
namespace math {

// ---- polynomial sine / log used by the pass prologue and by Quaternion ------------------------------
// reference: Core/Math/Transcendental.cpp:26-49 (scalar Sin truncates in its range reduction),
// :51-76 (the vector Sin used by SinCos rounds to nearest and fuses its Horner steps), :194-214 (FastLog)
namespace {
const float kS0 = 9.9999970197e-01f, kS1 = -1.6666577756e-01f, kS2 = 8.3325579762e-03f;
const float kS3 = -1.9812576647e-04f, kS4 = 2.7040521217e-06f, kS5 = -2.0532988642e-08f;

inline float flipSign(float y, int32 i)
{
    uint32 u; memcpy(&u, &y, 4); u ^= ((uint32)i << 31); memcpy(&y, &u, 4); return y;
}
inline float sinLaneRN(float a)
{
    const int32 i = (int32)lrintf(a * (1.0f / RT_PI));
    const float x = fmaf(-(float)i, RT_PI, a);
    const float x2 = x * x;
    float y = fmaf(kS5, x2, kS4);
    y = fmaf(y, x2, kS3);
    y = fmaf(y, x2, kS2);
    y = fmaf(y, x2, kS1);
    y = fmaf(y, x2, kS0);
    y *= x;
    return flipSign(y, i);
}
} // namespace

float Sin(float x)
{
    const int32 i = static_cast<int32>(x * (1.0f / RT_PI));
    x -= static_cast<float>(i) * RT_PI;
    const float x2 = x * x;
    float y = x * (kS0 + x2 * (kS1 + x2 * (kS2 + x2 * (kS3 + x2 * (kS4 + x2 * kS5)))));
    return (i & 1) ? -y : y;
}

float Cos(float x) { return Sin(x + RT_PI / 2.0f); }

Vector4 SinCos(float x) { return Vector4(sinLaneRN(x + 0.0f), sinLaneRN(x + RT_PI / 2.0f), 0.0f, 0.0f); }

float FastLog(float x)
{
    int32 xi; memcpy(&xi, &x, 4);
    const int32 e = (xi - 0x3f2aaaab) & 0xff800000;
    const int32 mi = xi - e;
    float m; memcpy(&m, &mi, 4);
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

Vector4 GetFloatNormal2(const Float2 u)
{
    return sqrtf(-2.0f * FastLog(u.x)) * SinCos(2.0f * RT_PI * u.y);
}

// ---- Matrix4 -------------------------------------------------------------------------------------------
Matrix4 Matrix4::MakeTranslation(const Vector4& pos)
{
    Matrix4 m = Identity();
    m.rows[3] = Vector4(pos.x, pos.y, pos.z, 1.0f);
    return m;
}

Matrix4 Matrix4::MakeScaling(const Vector4& scale)
{
    Matrix4 m = Identity();
    m.rows[0] *= scale.x; m.rows[1] *= scale.y; m.rows[2] *= scale.z;
    return m;
}

Matrix4 Matrix4::operator*(const Matrix4& b) const
{
    Matrix4 r;
    for (int i = 0; i < 4; ++i)
    {
        const Vector4& a = rows[i];
        r.rows[i] = b.rows[0] * a.x + b.rows[1] * a.y + b.rows[2] * a.z + b.rows[3] * a.w;
    }
    return r;
}

// General inverse as adjugate / determinant.  The inverse is an INPUT of the device path (RtObject::invTransform,
// RtLight::invTransform), so a render only matches the reference's bit for bit if every rounding of it does: the reference
// (Core/Math/Matrix4.cpp:123-248) rounds each cofactor as six triple products (e * p) * s summed left to right, and the 3x3
// minor is expanded along its first remaining column, rows ascending.  Written as that expansion rather than as the usual
// 2x2 sub-determinant sharing (which is as accurate but rounds differently: tests/golden/host_inverse.kat, tools/reference_fuzz.py).
Matrix4 Matrix4::Inverse() const
{
    const float* m = &rows[0].x;
    float adj[16];
    for (int r = 0; r < 4; ++r)
    {
        for (int c = 0; c < 4; ++c)
        {
            // adj[r][c] = (-1)^(r+c) * minor with row c and column r struck out
            int rr[3], cc[3];
            for (int i = 0, k = 0; i < 4; ++i) if (i != c) rr[k++] = i;
            for (int i = 0, k = 0; i < 4; ++i) if (i != r) cc[k++] = i;
            float sign = ((r + c) & 1) ? -1.0f : 1.0f;
            float sum = 0.0f;
            for (int a = 0; a < 3; ++a)
            {
                const int lo = rr[a == 0 ? 1 : 0], hi = rr[a == 2 ? 1 : 2];   // the two rows left, ascending
                const float e = sign * m[4 * rr[a] + cc[0]];
                const float t0 = (e * m[4 * lo + cc[1]]) * m[4 * hi + cc[2]];
                const float t1 = (e * m[4 * lo + cc[2]]) * m[4 * hi + cc[1]];
                sum = (a == 0) ? t0 - t1 : (sum + t0) - t1;
                sign = -sign;
            }
            adj[4 * r + c] = sum;
        }
    }
    float det = m[0] * adj[0] + m[1] * adj[4] + m[2] * adj[8] + m[3] * adj[12];
    det = 1.0f / det;
    Matrix4 out;
    float* o = &out.rows[0].x;
    for (int i = 0; i < 16; ++i) o[i] = adj[i] * det;
    return out;
}

Box Matrix4::TransformBox(const Box& box) const
{
    // per-axis extremes of the transformed corners (reference: Core/Math/Matrix4.cpp:253-268)
    const Vector4 xa = rows[0] * box.min.x, xb = rows[0] * box.max.x;
    const Vector4 ya = rows[1] * box.min.y, yb = rows[1] * box.max.y;
    const Vector4 za = rows[2] * box.min.z, zb = rows[2] * box.max.z;
    return Box(Vector4::Min(xa, xb) + Vector4::Min(ya, yb) + Vector4::Min(za, zb) + rows[3],
               Vector4::Max(xa, xb) + Vector4::Max(ya, yb) + Vector4::Max(za, zb) + rows[3]);
}

// ---- Quaternion ----------------------------------------------------------------------------------------
Quaternion Quaternion::FromAxisAndAngle(const Vector4& axis, float angle)
{
    angle *= 0.5f;
    Quaternion r(axis * Sin(angle));
    r.q.w = Cos(angle);
    return r;
}
Quaternion Quaternion::RotationX(float angle) { angle *= 0.5f; return Quaternion(Sin(angle), 0.0f, 0.0f, Cos(angle)); }
Quaternion Quaternion::RotationY(float angle) { angle *= 0.5f; return Quaternion(0.0f, Sin(angle), 0.0f, Cos(angle)); }
Quaternion Quaternion::RotationZ(float angle) { angle *= 0.5f; return Quaternion(0.0f, 0.0f, Sin(angle), Cos(angle)); }

Quaternion Quaternion::FromEulerAngles(const Float3& angles)
{
    const float pitch = angles.x * 0.5f, yaw = angles.y * 0.5f, roll = angles.z * 0.5f;
    const float cy = Cos(yaw), sy = Sin(yaw), cr = Cos(roll), sr = Sin(roll), cp = Cos(pitch), sp = Sin(pitch);
    return Quaternion(cy * cr * sp + sy * sr * cp,
                      sy * cr * cp - cy * sr * sp,
                      cy * sr * cp - sy * cr * sp,
                      cy * cr * cp + sy * sr * sp);
}

Quaternion Quaternion::operator*(const Quaternion& b) const
{
    const Vector4& a = q;
    return Quaternion(a.w * b.q.x - a.z * b.q.y + a.y * b.q.z + a.x * b.q.w,
                      a.w * b.q.y - a.x * b.q.z + a.z * b.q.x + a.y * b.q.w,
                      a.w * b.q.z - a.y * b.q.x + a.x * b.q.y + a.z * b.q.w,
                      a.w * b.q.w - a.z * b.q.z - a.y * b.q.y - a.x * b.q.x);
}

Quaternion Quaternion::Normalized() const
{
    const float l = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Quaternion(q.x / l, q.y / l, q.z / l, q.w / l);
}

Vector4 Quaternion::GetAxisX() const { return Vector4(1.0f - 2.0f * (q.y * q.y + q.z * q.z), 2.0f * (q.x * q.y + q.w * q.z), 2.0f * (q.x * q.z - q.w * q.y), 0.0f); }
Vector4 Quaternion::GetAxisY() const { return Vector4(2.0f * (q.x * q.y - q.w * q.z), 1.0f - 2.0f * (q.x * q.x + q.z * q.z), 2.0f * (q.y * q.z + q.w * q.x), 0.0f); }
Vector4 Quaternion::GetAxisZ() const { return Vector4(2.0f * (q.x * q.z + q.w * q.y), 2.0f * (q.y * q.z - q.w * q.x), 1.0f - 2.0f * (q.x * q.x + q.y * q.y), 0.0f); }

Matrix4 Quaternion::ToMatrix4() const
{
    return Matrix4(GetAxisX(), GetAxisY(), GetAxisZ(), VECTOR_W);
}

// ---- Entropy / Random ------------------------------------------------------------------------------------
namespace {
std::mutex gEntropyMutex;
bool gEntropySeeded = false;
bool gEntropyEnvChecked = false;
uint64 gEntropyState = 0;

inline uint64 splitmix64(uint64& s)
{
    uint64 z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
inline uint64 rotl64(uint64 x, int k) { return (x << k) | (x >> (64 - k)); }
} // namespace

void Entropy::SetGlobalSeed(uint64 seed)
{
    std::lock_guard<std::mutex> lock(gEntropyMutex);
    gEntropyEnvChecked = true;
    gEntropySeeded = seed != 0;
    gEntropyState = seed;
}

Entropy::Entropy() : mFd(-1)
{
    {
        std::lock_guard<std::mutex> lock(gEntropyMutex);
        if (!gEntropyEnvChecked)
        {
            gEntropyEnvChecked = true;
            if (const char* env = getenv("RT_SEED"))
            {
                gEntropyState = strtoull(env, nullptr, 0);
                gEntropySeeded = gEntropyState != 0;
            }
        }
    }
    mFd = open("/dev/urandom", O_RDONLY);
}

Entropy::~Entropy() { if (mFd >= 0) close(mFd); }

uint32 Entropy::GetInt()
{
    {
        std::lock_guard<std::mutex> lock(gEntropyMutex);
        if (gEntropySeeded) return (uint32)(splitmix64(gEntropyState) >> 32);
    }
    uint32 v = 0;
    if (mFd < 0 || read(mFd, &v, sizeof(v)) != (ssize_t)sizeof(v))
    {
        static uint64 fallback = 0x1234567887654321ULL;   // no entropy source: still return something usable
        v = (uint32)(splitmix64(fallback) >> 32);
    }
    return v;
}

Random::Random() { Reset(); }

void SetFlushDenormalsToZero(bool enable)
{
#if defined(__SSE__)
    unsigned int csr = __builtin_ia32_stmxcsr();
    csr = enable ? (csr | 0x8040u) : (csr & ~0x8040u);   // FTZ (bit 15) + DAZ (bit 6)
    __builtin_ia32_ldmxcsr(csr);
#else
    (void)enable;
#endif
}
bool GetFlushDenormalsToZero()
{
#if defined(__SSE__)
    return (__builtin_ia32_stmxcsr() & 0x8040u) == 0x8040u;
#else
    return true;
#endif
}

void Random::Reset()
{
    Entropy entropy;
    for (uint32 i = 0; i < 2; ++i)
    {
        mSeed[i] = ((uint64)entropy.GetInt() << 32) | (uint64)entropy.GetInt();
        for (uint32 l = 0; l < 2; ++l) mSeedSimd4[i][l] = ((uint64)entropy.GetInt() << 32) | (uint64)entropy.GetInt();
    }
}

void Random::Reset(uint64 seed)
{
    uint64 s = seed;
    mSeed[0] = splitmix64(s); mSeed[1] = splitmix64(s) | 1ULL;
    for (uint32 i = 0; i < 2; ++i) for (uint32 l = 0; l < 2; ++l) mSeedSimd4[i][l] = splitmix64(s) | 1ULL;
}

void Random::SetState(const uint64 scalarState[2], const uint64 simd4State[4])
{
    mSeed[0] = scalarState[0]; mSeed[1] = scalarState[1];
    mSeedSimd4[0][0] = simd4State[0]; mSeedSimd4[0][1] = simd4State[1];
    mSeedSimd4[1][0] = simd4State[2]; mSeedSimd4[1][1] = simd4State[3];
}

// xoroshiro128+ (reference: Core/Math/Random.cpp:33-47)
uint64 Random::GetLong()
{
    const uint64 s0 = mSeed[0];
    uint64 s1 = mSeed[1];
    const uint64 result = s0 + s1;
    s1 ^= s0;
    mSeed[0] = rotl64(s0, 24) ^ s1 ^ (s1 << 16);
    mSeed[1] = rotl64(s1, 37);
    return result;
}

uint32 Random::GetInt() { return static_cast<uint32>(GetLong()); }

float Random::GetFloat()
{
    const uint32 u = (GetInt() & 0x007fffffu) | 0x3f800000u;
    float f; memcpy(&f, &u, 4);
    return f - 1.0f;
}

double Random::GetDouble()
{
    return static_cast<double>(GetLong()) / static_cast<double>(std::numeric_limits<uint64>::max());
}

// two independent xorshift128+ 64-bit lanes (reference: Core/Math/Random.cpp:83-114)
void Random::GetIntVector4(uint64 out[2])
{
    for (uint32 l = 0; l < 2; ++l)
    {
        const uint64 s0 = mSeedSimd4[1][l];
        uint64 s1 = mSeedSimd4[0][l];
        out[l] = s0 + s1;
        s1 <<= 23;
        const uint64 t0 = s0 >> 5;
        const uint64 t1 = s1 >> 18;
        mSeedSimd4[0][l] = s0;
        mSeedSimd4[1][l] = (s0 ^ s1) ^ (t0 ^ t1);
    }
}

Vector4 Random::GetVector4()
{
    uint64 v[2];
    GetIntVector4(v);
    uint32 lanes[4] = { (uint32)v[0], (uint32)(v[0] >> 32), (uint32)v[1], (uint32)(v[1] >> 32) };
    Vector4 r;
    for (uint32 i = 0; i < 4; ++i)
    {
        const uint32 u = (lanes[i] & 0x007fffffu) | 0x3f800000u;
        float f; memcpy(&f, &u, 4);
        r[i] = f - 1.0f;
    }
    return r;
}

Float2 Random::GetFloat2() { return GetVector4().ToFloat2(); }
Float3 Random::GetFloat3() { return GetVector4().ToFloat3(); }

} // namespace math
} // namespace rt
