// dmath.hpp — fp32 vector math for the HIP kernels (and the host-side setup code).
//
// Parity contract: every routine performs the reference's operations in the reference's order
// (src/core/math.h, src/core/math.cpp) so accept/reject compares in the narrow phase see the same
// rounded inputs as the CPU path.  Build with -ffp-contract=off and IEEE div/sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt); never with fast-math.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#define MI_HD __host__ __device__ __forceinline__

namespace mi {

constexpr float kPi = 3.14159265359f;   // src/core/math.h:13
constexpr float kEps = 1e-6f;           // src/core/math.h:22
constexpr float kGravity = -9.81f;      // src/physics/physics.h:11

struct V3 {
    float x, y, z;
    MI_HD V3() : x(0.f), y(0.f), z(0.f) {}
    MI_HD explicit V3(float s) : x(s), y(s), z(s) {}
    MI_HD V3(float a, float b, float c) : x(a), y(b), z(c) {}
    MI_HD float get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    MI_HD void set(int i, float v) { if (i == 0) x = v; else if (i == 1) y = v; else z = v; }
};
struct P4 { float x, y, z, w; };  // plane (n, d)
struct Q4 {
    float x, y, z, w;
    MI_HD Q4() : x(0.f), y(0.f), z(0.f), w(1.f) {}
    MI_HD Q4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    MI_HD V3 v() const { return V3(x, y, z); }
};
// Row-major struct of the 3x3 entries; r(i) and c(j) give rows / columns.
struct M3 {
    float m00, m01, m02, m10, m11, m12, m20, m21, m22;
    MI_HD V3 r(int i) const { return i == 0 ? V3(m00, m01, m02) : (i == 1 ? V3(m10, m11, m12) : V3(m20, m21, m22)); }
    MI_HD V3 c(int j) const { return j == 0 ? V3(m00, m10, m20) : (j == 1 ? V3(m01, m11, m21) : V3(m02, m12, m22)); }
    MI_HD static M3 zero() { M3 m; m.m00 = m.m01 = m.m02 = m.m10 = m.m11 = m.m12 = m.m20 = m.m21 = m.m22 = 0.f; return m; }
    MI_HD static M3 identity() { M3 m = zero(); m.m00 = m.m11 = m.m22 = 1.f; return m; }
};

MI_HD float fminr(float a, float b) { return a < b ? a : b; }   // Windows min/max macros
MI_HD float fmaxr(float a, float b) { return a > b ? a : b; }
MI_HD float clampr(float v, float l, float u) { float r = fmaxr(l, v); r = fminr(u, r); return r; }  // math.h:30
MI_HD float clamp01(float v) { return clampr(v, 0.f, 1.f); }
MI_HD float lerpr(float l, float u, float t) { return l + t * (u - l); }

MI_HD V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
MI_HD V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
MI_HD V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
MI_HD V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
MI_HD V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
MI_HD V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
MI_HD V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
MI_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MI_HD V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MI_HD float sqlen(V3 a) { return dot(a, a); }
MI_HD float len(V3 a) { return sqrtf(sqlen(a)); }
MI_HD V3 normalize(V3 a) { float l = len(a); return a * (1.f / l); }                                   // math.h:599
MI_HD V3 noz(V3 a) { float sl = sqlen(a); return (sl < 1e-8f) ? V3() : (a * (1.f / sqrtf(sl))); }      // math.h:595
MI_HD V3 vabs(V3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
MI_HD V3 vmin(V3 a, V3 b) { return V3(fminr(a.x, b.x), fminr(a.y, b.y), fminr(a.z, b.z)); }
MI_HD V3 vmax(V3 a, V3 b) { return V3(fmaxr(a.x, b.x), fmaxr(a.y, b.y), fmaxr(a.z, b.z)); }
MI_HD V3 lerp(V3 l, V3 u, float t) { return l + t * (u - l); }
// vec4 dot in the reference is two _mm_hadd_ps: (x+y)+(z+w)  (src/core/simd.h:349)
MI_HD float dot4h(float ax, float ay, float az, float aw, float bx, float by, float bz, float bw) {
    return (ax * bx + ay * by) + (az * bz + aw * bw);
}

MI_HD Q4 conj(Q4 a) { return Q4(-a.x, -a.y, -a.z, a.w); }
MI_HD Q4 operator*(Q4 a, Q4 b) {  // math.h:627-633
    Q4 r;
    r.w = a.w * b.w - dot(a.v(), b.v());
    V3 v = a.v() * b.w + b.v() * a.w + cross(a.v(), b.v());
    r.x = v.x; r.y = v.y; r.z = v.z;
    return r;
}
MI_HD V3 rotate(Q4 q, V3 v) {  // quat * vec3 as the sandwich product, math.h:642-646
    Q4 p(v.x, v.y, v.z, 0.f);
    return (q * p * conj(q)).v();
}
MI_HD Q4 normalize(Q4 a) {
    float l = sqrtf(dot4h(a.x, a.y, a.z, a.w, a.x, a.y, a.z, a.w));
    float s = 1.f / l;
    return Q4(a.x * s, a.y * s, a.z * s, a.w * s);
}
MI_HD bool isIdentity(Q4 q) { return q.x == 0.f && q.y == 0.f && q.z == 0.f && q.w == 1.f; }

MI_HD M3 mul(const M3& a, const M3& b) {  // math.cpp:103-118
    M3 r;
    r.m00 = dot(a.r(0), b.c(0)); r.m01 = dot(a.r(0), b.c(1)); r.m02 = dot(a.r(0), b.c(2));
    r.m10 = dot(a.r(1), b.c(0)); r.m11 = dot(a.r(1), b.c(1)); r.m12 = dot(a.r(1), b.c(2));
    r.m20 = dot(a.r(2), b.c(0)); r.m21 = dot(a.r(2), b.c(1)); r.m22 = dot(a.r(2), b.c(2));
    return r;
}
MI_HD V3 mul(const M3& a, V3 b) { return V3(dot(a.r(0), b), dot(a.r(1), b), dot(a.r(2), b)); }
MI_HD M3 transpose(const M3& a) {
    M3 r;
    r.m00 = a.m00; r.m01 = a.m10; r.m02 = a.m20;
    r.m10 = a.m01; r.m11 = a.m11; r.m12 = a.m21;
    r.m20 = a.m02; r.m21 = a.m12; r.m22 = a.m22;
    return r;
}
MI_HD M3 scale(const M3& a, float s) {
    M3 r;
    r.m00 = a.m00 * s; r.m01 = a.m01 * s; r.m02 = a.m02 * s;
    r.m10 = a.m10 * s; r.m11 = a.m11 * s; r.m12 = a.m12 * s;
    r.m20 = a.m20 * s; r.m21 = a.m21 * s; r.m22 = a.m22 * s;
    return r;
}
MI_HD M3 add(const M3& a, const M3& b) {
    M3 r;
    r.m00 = a.m00 + b.m00; r.m01 = a.m01 + b.m01; r.m02 = a.m02 + b.m02;
    r.m10 = a.m10 + b.m10; r.m11 = a.m11 + b.m11; r.m12 = a.m12 + b.m12;
    r.m20 = a.m20 + b.m20; r.m21 = a.m21 + b.m21; r.m22 = a.m22 + b.m22;
    return r;
}
MI_HD M3 sub(const M3& a, const M3& b) {
    M3 r;
    r.m00 = a.m00 - b.m00; r.m01 = a.m01 - b.m01; r.m02 = a.m02 - b.m02;
    r.m10 = a.m10 - b.m10; r.m11 = a.m11 - b.m11; r.m12 = a.m12 - b.m12;
    r.m20 = a.m20 - b.m20; r.m21 = a.m21 - b.m21; r.m22 = a.m22 - b.m22;
    return r;
}
MI_HD float det(const M3& m) {  // math.cpp:443-448
    return m.m00 * (m.m11 * m.m22 - m.m21 * m.m12) - m.m01 * (m.m10 * m.m22 - m.m20 * m.m12) + m.m02 * (m.m10 * m.m21 - m.m20 * m.m11);
}
MI_HD M3 invert(const M3& m) {  // math.cpp:276-318
    M3 inv;
    inv.m00 = m.m11 * m.m22 - m.m21 * m.m12;
    inv.m01 = m.m02 * m.m21 - m.m22 * m.m01;
    inv.m02 = m.m01 * m.m12 - m.m11 * m.m02;
    inv.m10 = m.m12 * m.m20 - m.m22 * m.m10;
    inv.m11 = m.m00 * m.m22 - m.m20 * m.m02;
    inv.m12 = m.m02 * m.m10 - m.m12 * m.m00;
    inv.m20 = m.m10 * m.m21 - m.m20 * m.m11;
    inv.m21 = m.m01 * m.m20 - m.m21 * m.m00;
    inv.m22 = m.m00 * m.m11 - m.m10 * m.m01;
    float d = det(m);
    if (d == 0.f) return M3::zero();
    d = 1.f / d;
    return scale(inv, d);
}
MI_HD M3 outer(V3 a, V3 b) {  // math.cpp:778-795 (a * b^T)
    M3 r;
    r.m00 = a.x * b.x; r.m10 = a.y * b.x; r.m20 = a.z * b.x;
    r.m01 = a.x * b.y; r.m11 = a.y * b.y; r.m21 = a.z * b.y;
    r.m02 = a.x * b.z; r.m12 = a.y * b.z; r.m22 = a.z * b.z;
    return r;
}
MI_HD M3 skew(V3 r) {  // math.cpp:797-811
    M3 m;
    m.m00 = 0.f; m.m01 = -r.z; m.m02 = r.y;
    m.m10 = r.z; m.m11 = 0.f; m.m12 = -r.x;
    m.m20 = -r.y; m.m21 = r.x; m.m22 = 0.f;
    return m;
}
MI_HD M3 quatToMat(Q4 q) {  // math.cpp:644-674
    if (q.w == 1.f) return M3::identity();
    float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
    float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    M3 r;
    r.m00 = 1.f - 2.f * (qyy + qzz);
    r.m10 = 2.f * (qxy + qwz);
    r.m20 = 2.f * (qxz - qwy);
    r.m01 = 2.f * (qxy - qwz);
    r.m11 = 1.f - 2.f * (qxx + qzz);
    r.m21 = 2.f * (qyz + qwx);
    r.m02 = 2.f * (qxz + qwy);
    r.m12 = 2.f * (qyz - qwx);
    r.m22 = 1.f - 2.f * (qxx + qyy);
    return r;
}
MI_HD V3 solve3(const M3& A, V3 b) {  // math.cpp:1356-1371 (Cramer; det == 0 -> 0)
    V3 ex = A.c(0), ey = A.c(1), ez = A.c(2);
    float d = dot(ex, cross(ey, ez));
    if (d != 0.f) d = 1.f / d;
    V3 x;
    x.x = d * dot(b, cross(ey, ez));
    x.y = d * dot(ex, cross(b, ez));
    x.z = d * dot(ex, cross(ey, b));
    return x;
}
MI_HD void solve2(float a11, float a12, float a21, float a22, float bx, float by, float& ox, float& oy) {  // math.cpp:1342-1354
    float d = a11 * a22 - a12 * a21;
    if (d != 0.f) d = 1.f / d;
    ox = d * (a22 * bx - a12 * by);
    oy = d * (a11 * by - a21 * bx);
}
MI_HD V3 tangentOf(V3 n) {  // getTangent, math.cpp:1416-1420
    V3 t = (fabsf(n.x) >= 0.57735f) ? V3(n.y, -n.x, 0.f) : V3(0.f, n.z, -n.y);
    return normalize(t);
}
MI_HD P4 makePlane(V3 point, V3 normal) { float d = -dot(normal, point); return P4{normal.x, normal.y, normal.z, d}; }  // bounding_volumes.h:166
MI_HD float planeDist(V3 p, P4 pl) { return dot4h(p.x, p.y, p.z, 1.f, pl.x, pl.y, pl.z, pl.w); }                      // bounding_volumes.h:296

// ---- deterministic transcendentals -------------------------------------------------------------
// The reference calls the MSVC CRT (acos, atan2, sin, cos).  No two libms agree to the last ulp, so
// the GPU path uses fixed +,-,*,/,sqrt sequences in double (error < 1e-13, i.e. correctly rounded
// to fp32 except in rare ties); tests/test_transcendentals.py bounds them against libm.
MI_HD double atanPoly(double t) {
    double u = t * t;
    double q = 3.07024230903805012e-02;
    q = q * u + -5.87725109466260137e-02;
    q = q * u + 7.56446973448029469e-02;
    q = q * u + -9.07850346883833786e-02;
    q = q * u + 1.11103940245759952e-01;
    q = q * u + -1.42856908172007746e-01;
    q = q * u + 1.99999996133686908e-01;
    q = q * u + -3.33333333308731938e-01;
    q = q * u + 9.99999999999974576e-01;
    return q * t;
}
MI_HD double atan01(double a) {
    if (a > 0.41421356237309503) return 0.78539816339744828 + atanPoly((a - 1.0) / (a + 1.0));
    return atanPoly(a);
}
MI_HD float detAtan2(float y, float x) {
    double ax = fabs((double)x), ay = fabs((double)y);
    double r;
    if (ax == 0.0 && ay == 0.0) r = 0.0;
    else if (ay <= ax) r = atan01(ay / ax);
    else r = 1.5707963267948966 - atan01(ax / ay);
    if (x < 0.f) r = 3.141592653589793 - r;
    if (y < 0.f) r = -r;
    return (float)r;
}
MI_HD float detAcos(float x) {
    float c = clampr(x, -1.f, 1.f);
    double d = (double)c;
    double s = sqrt((1.0 - d) * (1.0 + d));
    return detAtan2((float)s, c);
}
MI_HD double sinCore(double x) {
    double x2 = x * x;
    double p = -2.5052108385441720e-08;
    p = p * x2 + 2.7557319223985893e-06;
    p = p * x2 + -1.9841269841269841e-04;
    p = p * x2 + 8.3333333333333332e-03;
    p = p * x2 + -1.6666666666666666e-01;
    return x + x * x2 * p;
}
MI_HD double cosCore(double x) {
    double x2 = x * x;
    double p = 2.0876756987868100e-09;
    p = p * x2 + -2.7557319223985888e-07;
    p = p * x2 + 2.4801587301587302e-05;
    p = p * x2 + -1.3888888888888889e-03;
    p = p * x2 + 4.1666666666666664e-02;
    p = p * x2 + -0.5;
    return 1.0 + x2 * p;
}
MI_HD void detSinCos(float xf, float& so, float& co) {
    double x = (double)xf;
    double q = floor(x * 0.6366197723675814 + 0.5);
    double r = x - q * 1.5707963267948966;
    r = r - q * 6.123233995736766e-17;
    long long k = (long long)q;
    double sr = sinCore(r), cr = cosCore(r), s, c;
    switch (k & 3) {
        case 0: s = sr; c = cr; break;
        case 1: s = cr; c = -sr; break;
        case 2: s = -sr; c = -cr; break;
        default: s = -cr; c = sr; break;
    }
    so = (float)s; co = (float)c;
}

MI_HD Q4 rotateFromTo(V3 from_, V3 to_) {  // math.cpp:538-575
    V3 from = normalize(from_), to = normalize(to_);
    float d = dot(from, to);
    if (d >= 1.f) return Q4(0.f, 0.f, 0.f, 1.f);
    Q4 q;
    if (d < (1e-6f - 1.f)) {
        V3 axis = cross(V3(1.f, 0.f, 0.f), from);
        if (sqlen(axis) == 0.f) axis = cross(V3(0.f, 1.f, 0.f), from);
        axis = normalize(axis);
        float s, c;
        detSinCos(kPi * 0.5f, s, c);
        q = normalize(Q4(axis.x * s, axis.y * s, axis.z * s, c));
    } else {
        float s = sqrtf((1.f + d) * 2.f);
        float invs = 1.f / s;
        V3 c = cross(from, to);
        q.x = c.x * invs; q.y = c.y * invs; q.z = c.z * invs; q.w = s * 0.5f;
        q = normalize(q);
    }
    return q;
}
MI_HD void axisRotation(Q4 q, V3& axis, float& angle) {  // getAxisRotation, math.cpp:577-593
    float sl = sqlen(q.v());
    if (sl > 0.f) {
        angle = 2.f * detAcos(q.w);
        float il = 1.f / sqrtf(sl);
        axis = q.v() * il;
    } else {
        angle = 0.f;
        axis = V3(1.f, 0.f, 0.f);
    }
}

MI_HD uint32_t hash32(uint32_t m) {  // bijection on 32 bits -> unique colouring priorities
    uint32_t h = m * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h;
}

// float4 helpers
MI_HD V3 xyz(const float4& f) { return V3(f.x, f.y, f.z); }
MI_HD float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
MI_HD Q4 toQ(const float4& f) { return Q4(f.x, f.y, f.z, f.w); }
MI_HD float4 fromQ(Q4 q) { return make_float4(q.x, q.y, q.z, q.w); }

}  // namespace mi
