// ora_math.h — TEST INFRASTRUCTURE (CPU oracle).  Not part of the product; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load anything under oracle/.
//
// Strict-IEEE fp32 restatement of the vector math the reference physics path uses.  Every routine
// keeps the reference's operation ORDER (so accept/reject compares see identically rounded
// inputs); compile with -ffp-contract=off -fno-fast-math.  Citations are relative to the
// reference repo root.
#pragma once
#include <cmath>
#include <cfloat>
#include <cstdint>

namespace ora {

static const float kPi = 3.14159265359f;  // src/core/math.h:13
static const float kEps = 1e-6f;          // src/core/math.h:22

struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float v) : x(v), y(v), z(v) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct vec4 { float x, y, z, w; };
struct quat {
    float x, y, z, w;
    quat() : x(0), y(0), z(0), w(1) {}
    quat(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    vec3 v() const { return vec3(x, y, z); }
};
// Column-major storage like the reference (src/core/math.h:390-397): m[col*3+row].
struct mat3 {
    float m00, m10, m20, m01, m11, m21, m02, m12, m22;
    static mat3 zero() { mat3 r; r.m00 = r.m10 = r.m20 = r.m01 = r.m11 = r.m21 = r.m02 = r.m12 = r.m22 = 0.f; return r; }
    static mat3 identity() { mat3 r = zero(); r.m00 = r.m11 = r.m22 = 1.f; return r; }
    float* data() { return &m00; }
    const float* data() const { return &m00; }
};

// src/core/math.h:532-548
static inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator*(vec3 a, float b) { return vec3(a.x * b, a.y * b, a.z * b); }
static inline vec3 operator*(float a, vec3 b) { return b * a; }
static inline vec3 operator/(vec3 a, float b) { return vec3(a.x / b, a.y / b, a.z / b); }
static inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
static inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
static inline vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
static inline vec3& operator*=(vec3& a, float b) { a = a * b; return a; }
static inline vec3& operator/=(vec3& a, float b) { a = a / b; return a; }

static inline float fmin2(float a, float b) { return a < b ? a : b; }
static inline float fmax2(float a, float b) { return a > b ? a : b; }
static inline float clampf(float v, float l, float u) { float r = fmax2(l, v); r = fmin2(u, r); return r; }  // math.h:30
static inline float clamp01(float v) { return clampf(v, 0.f, 1.f); }
static inline float lerpf(float l, float u, float t) { return l + t * (u - l); }  // math.h:27

// src/core/math.h:580-600
static inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float squaredLength(vec3 a) { return dot(a, a); }
static inline float length(vec3 a) { return std::sqrt(squaredLength(a)); }
static inline vec3 noz(vec3 a) { float sl = squaredLength(a); return (sl < 1e-8f) ? vec3(0.f, 0.f, 0.f) : (a * (1.f / std::sqrt(sl))); }
static inline vec3 normalize(vec3 a) { float l = length(a); return a * (1.f / l); }
static inline vec3 vabs(vec3 a) { return vec3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }
static inline vec3 vmin(vec3 a, vec3 b) { return vec3(fmin2(a.x, b.x), fmin2(a.y, b.y), fmin2(a.z, b.z)); }
static inline vec3 vmax(vec3 a, vec3 b) { return vec3(fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z)); }
static inline vec3 lerp(vec3 l, vec3 u, float t) { return l + t * (u - l); }  // math.h:671
static inline bool operator==(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// vec4 dot goes through _mm_hadd_ps twice: (x+y)+(z+w)  (src/core/simd.h:349, math.h:582).
static inline float dot4(float ax, float ay, float az, float aw, float bx, float by, float bz, float bw) {
    return (ax * bx + ay * by) + (az * bz + aw * bw);
}

// src/core/math.h:622-646
static inline quat conjugate(quat a) { return quat(-a.x, -a.y, -a.z, a.w); }
static inline quat operator*(quat a, quat b) {
    quat r;
    r.w = a.w * b.w - dot(a.v(), b.v());
    vec3 v = a.v() * b.w + b.v() * a.w + cross(a.v(), b.v());
    r.x = v.x; r.y = v.y; r.z = v.z;
    return r;
}
static inline quat operator*(quat q, float s) { return quat(q.x * s, q.y * s, q.z * s, q.w * s); }
static inline quat operator+(quat a, quat b) { return quat(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline vec3 operator*(quat q, vec3 v) {
    quat p(v.x, v.y, v.z, 0.f);
    return (q * p * conjugate(q)).v();
}
static inline quat normalize(quat a) {
    float l = std::sqrt(dot4(a.x, a.y, a.z, a.w, a.x, a.y, a.z, a.w));
    float s = 1.f / l;
    return quat(a.x * s, a.y * s, a.z * s, a.w * s);
}
static inline bool operator==(quat a, quat b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

static inline vec3 row(const mat3& a, int r) { const float* m = a.data(); return vec3(m[r], m[3 + r], m[6 + r]); }
static inline vec3 col(const mat3& a, int c) { const float* m = a.data(); return vec3(m[c * 3], m[c * 3 + 1], m[c * 3 + 2]); }

// src/core/math.cpp:103-118
static inline mat3 operator*(const mat3& a, const mat3& b) {
    vec3 r0 = row(a, 0), r1 = row(a, 1), r2 = row(a, 2);
    vec3 c0 = col(b, 0), c1 = col(b, 1), c2 = col(b, 2);
    mat3 r;
    r.m00 = dot(r0, c0); r.m01 = dot(r0, c1); r.m02 = dot(r0, c2);
    r.m10 = dot(r1, c0); r.m11 = dot(r1, c1); r.m12 = dot(r1, c2);
    r.m20 = dot(r2, c0); r.m21 = dot(r2, c1); r.m22 = dot(r2, c2);
    return r;
}
static inline mat3 operator+(const mat3& a, const mat3& b) { mat3 r; for (int i = 0; i < 9; ++i) r.data()[i] = a.data()[i] + b.data()[i]; return r; }
static inline mat3 operator-(const mat3& a, const mat3& b) { mat3 r; for (int i = 0; i < 9; ++i) r.data()[i] = a.data()[i] - b.data()[i]; return r; }
static inline mat3 operator*(const mat3& a, float b) { mat3 r; for (int i = 0; i < 9; ++i) r.data()[i] = a.data()[i] * b; return r; }
static inline mat3 operator*(float b, const mat3& a) { return a * b; }
static inline vec3 operator*(const mat3& a, vec3 b) { return vec3(dot(row(a, 0), b), dot(row(a, 1), b), dot(row(a, 2), b)); }  // math.h:658
static inline mat3 transpose(const mat3& a) {  // math.cpp:241-248
    mat3 r;
    r.m00 = a.m00; r.m01 = a.m10; r.m02 = a.m20;
    r.m10 = a.m01; r.m11 = a.m11; r.m12 = a.m21;
    r.m20 = a.m02; r.m21 = a.m12; r.m22 = a.m22;
    return r;
}
static inline float trace(const mat3& a) { return a.m00 + a.m11 + a.m22; }
static inline float determinant(const mat3& m) {  // math.cpp:443-448
    return m.m00 * (m.m11 * m.m22 - m.m21 * m.m12) - m.m01 * (m.m10 * m.m22 - m.m20 * m.m12) + m.m02 * (m.m10 * m.m21 - m.m20 * m.m11);
}
mat3 invert(const mat3& m);
static inline mat3 outerProduct(vec3 a, vec3 b) {  // math.cpp:792-811: result = a * b^T
    mat3 r;
    r.m00 = a.x * b.x; r.m10 = a.y * b.x; r.m20 = a.z * b.x;
    r.m01 = a.x * b.y; r.m11 = a.y * b.y; r.m21 = a.z * b.y;
    r.m02 = a.x * b.z; r.m12 = a.y * b.z; r.m22 = a.z * b.z;
    return r;
}
static inline mat3 getSkewMatrix(vec3 r) {  // math.cpp:797-811
    mat3 m;
    m.m00 = 0.f; m.m01 = -r.z; m.m02 = r.y;
    m.m10 = r.z; m.m11 = 0.f; m.m12 = -r.x;
    m.m20 = -r.y; m.m21 = r.x; m.m22 = 0.f;
    return m;
}

// src/core/math.cpp:644-674
static inline mat3 quaternionToMat3(quat q) {
    if (q.w == 1.f) return mat3::identity();
    float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
    float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    mat3 r;
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

quat rotateFromTo(vec3 from, vec3 to);                 // math.cpp:538-575
void getAxisRotation(quat q, vec3& axis, float& angle); // math.cpp:577-593

// src/core/math.cpp:1342-1371 (Cramer; det == 0 -> x = 0)
static inline void solveLinearSystem2(float a11, float a12, float a21, float a22, float bx, float by, float& ox, float& oy) {
    float det = a11 * a22 - a12 * a21;
    if (det != 0.f) det = 1.f / det;
    ox = det * (a22 * bx - a12 * by);
    oy = det * (a11 * by - a21 * bx);
}
static inline vec3 solveLinearSystem(const mat3& A, vec3 b) {
    vec3 ex(A.m00, A.m10, A.m20), ey(A.m01, A.m11, A.m21), ez(A.m02, A.m12, A.m22);
    float det = dot(ex, cross(ey, ez));
    if (det != 0.f) det = 1.f / det;
    vec3 x;
    x.x = det * dot(b, cross(ey, ez));
    x.y = det * dot(ex, cross(b, ez));
    x.z = det * dot(ex, cross(ey, b));
    return x;
}

// src/core/math.cpp:1416-1427
static inline vec3 getTangent(vec3 n) {
    vec3 t = (std::fabs(n.x) >= 0.57735f) ? vec3(n.y, -n.x, 0.f) : vec3(0.f, n.z, -n.y);
    return normalize(t);
}
static inline void getTangents(vec3 n, vec3& t, vec3& b) { t = getTangent(n); b = cross(n, t); }

// plane = (normal, d); src/physics/bounding_volumes.h:166-170, 296-299 (vec4 dot => hadd order)
static inline vec4 createPlane(vec3 point, vec3 normal) { float d = -dot(normal, point); return vec4{normal.x, normal.y, normal.z, d}; }
static inline float signedDistanceToPlane(vec3 p, vec4 pl) { return dot4(p.x, p.y, p.z, 1.f, pl.x, pl.y, pl.z, pl.w); }

// Deterministic transcendental replacements (see DESIGN.md "transcendentals"): the reference calls
// the MSVC CRT acos/atan2; CPU libm and the GPU's ocml differ from it (and from each other) in the
// last ulp, so the oracle and the HIP kernels both use these fixed +,-,*,/,sqrt sequences.
float det_atan2f(float y, float x);
float det_acosf(float x);
float det_sinf(float x);
float det_cosf(float x);

}  // namespace ora
