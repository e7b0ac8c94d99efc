// narrow.hpp — device-side shape-pair contact generation (one pair per lane).
//
// Behavioural spec: the scalar intersection() routines of src/physics/collision_narrow.cpp
// (normal from A to B, depth >= 0, contact = midpoint, <= 4 points per manifold, OBB-OBB by
// 15-axis SAT + Sutherland-Hodgman clipping + 4-point reduction).  Operation order is kept so
// contact COUNTS are bit-exact against the CPU oracle.
#pragma once
#include "dmath.hpp"

namespace mi {

enum : int { T_SPHERE = 0, T_CAPSULE = 1, T_CYLINDER = 2, T_AABB = 3, T_OBB = 4, T_HULL = 5 };

// World-space shape, 12 floats (three float4 rows in HBM):
//   sphere   r0 = (center, radius)
//   capsule  r0 = (A, radius) r1 = (B, -)          (cylinder identical)
//   aabb     r0 = (min, -)    r1 = (max, -)
//   obb      r0 = (center, -) r1 = (radius, -) r2 = quat
//   hull     r0 = (position, geometry id bits) r2 = quat
struct Shape {
    int type;
    V3 a, b;
    float radius;
    Q4 rot;
    uint32_t hull;
};

struct Manifold {
    V3 p[4];
    float d[4];
    V3 n;
    uint32_t count;
};

struct ClipVert { V3 v; float depth; };
// Clip polygon in private memory (GJK kernel: segment clips) ...
struct ClipPoly {
    ClipVert pt[16]; uint32_t n;
    __device__ __forceinline__ ClipVert get(uint32_t i) const { return pt[i]; }
    __device__ __forceinline__ void put(uint32_t i, const ClipVert& v) { pt[i] = v; }
};
// ... and in LDS for the primitive kernel: dynamically indexed private arrays live in scratch (HBM-backed, ~1 us per
// dependent access); a [vertex][lane] LDS plane keeps the Sutherland-Hodgman ping-pong on chip.  A quad clipped by
// four planes has at most 8 vertices.
constexpr uint32_t kLdsPolyVerts = 8;
constexpr uint32_t kLdsPolyStride = 256;   // lanes per workgroup of k_narrow
struct LdsPoly {
    float4* p; uint32_t n;
    __device__ __forceinline__ ClipVert get(uint32_t i) const { float4 f = p[i * kLdsPolyStride]; ClipVert c; c.v = V3(f.x, f.y, f.z); c.depth = f.w; return c; }
    __device__ __forceinline__ void put(uint32_t i, const ClipVert& c) { p[i * kLdsPolyStride] = make_float4(c.v.x, c.v.y, c.v.z, c.depth); }
};

__device__ __forceinline__ void setc(Manifold& m, uint32_t i, V3 p, float d) { m.p[i] = p; m.d[i] = d; }

// 4-point reduction (collision_narrow.cpp:56-146)
template <class Poly>
__device__ inline void reduceManifold(const Poly& v, uint32_t n, V3 normal, Manifold& out) {
    if (n > 4) {
        V3 searchDir = tangentOf(normal);
        float best = dot(searchDir, v.get(0).v);
        uint32_t ri = 0;
        for (uint32_t i = 1; i < n; ++i) { float dd = dot(searchDir, v.get(i).v); if (dd > best) { ri = i; best = dd; } }
        { ClipVert c = v.get(ri); setc(out, 0, c.v, c.depth); }
        best = 0.f; ri = 0;
        for (uint32_t i = 0; i < n; ++i) { float sq = sqlen(v.get(i).v - out.p[0]); if (sq > best) { ri = i; best = sq; } }
        { ClipVert c = v.get(ri); setc(out, 1, c.v, c.depth); }
        float bestArea = 0.f; ri = 0;
        for (uint32_t i = 0; i < n; ++i) {
            V3 vi = v.get(i).v;
            V3 qa = out.p[0] - vi, qb = out.p[1] - vi;
            float area = 0.5f * dot(cross(qa, qb), normal);
            if (area > bestArea) { ri = i; bestArea = area; }
        }
        { ClipVert c = v.get(ri); setc(out, 2, c.v, c.depth); }
        bestArea = 0.f; ri = 0;
        for (uint32_t i = 0; i < n; ++i) {
            V3 vi = v.get(i).v;
            V3 qa = out.p[0] - vi, qb = out.p[1] - vi, qc = out.p[2] - vi;
            float a1 = 0.5f * dot(cross(qa, qb), normal);
            float a2 = 0.5f * dot(cross(qb, qc), normal);
            float a3 = 0.5f * dot(cross(qc, qa), normal);
            float area = fmaxr(fmaxr(a1, a2), a3);
            if (area > bestArea) { ri = i; bestArea = area; }
        }
        { ClipVert c = v.get(ri); setc(out, 3, c.v, c.depth); }
        out.count = 4;
    } else {
        out.count = n;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) if (i < n) { ClipVert c = v.get(i); setc(out, i, c.v, c.depth); }   // (static indices: the manifold stays in registers)
    }
}

__device__ __forceinline__ ClipVert clipEdge(ClipVert a, ClipVert b, float ad, float bd) {  // 154-163
    ad = fabsf(ad); bd = fabsf(bd);
    float total = ad + bd;
    float t = ad / total;
    ClipVert r; r.v = lerp(a.v, b.v, t); r.depth = lerpr(a.depth, b.depth, t);
    return r;
}

// Sutherland-Hodgman against planes pointing inside (166-222); result always lands in `output`.
template <class Poly>
__device__ inline void clipPolygon(Poly& input, const P4* planes, uint32_t numPlanes, Poly& output) {
    Poly* in = &input; Poly* out = &output;
    uint32_t ci = 0;
    for (; ci < numPlanes; ++ci) {
        P4 pl = planes[ci];
        out->n = 0;
        if (in->n == 0) break;
        ClipVert start = in->get(in->n - 1);
        for (uint32_t i = 0; i < in->n; ++i) {
            ClipVert end = in->get(i);
            float sd = planeDist(start.v, pl), ed = planeDist(end.v, pl);
            bool sIn = sd > 0.f, eIn = ed > 0.f;
            if (sIn && eIn) out->put(out->n++, end);
            else if (sIn) out->put(out->n++, clipEdge(start, end, sd, ed));
            else if (!sIn && eIn) { out->put(out->n++, clipEdge(start, end, sd, ed)); out->put(out->n++, end); }
            start = end;
        }
        Poly* tmp = in; in = out; out = tmp;
    }
    if (ci % 2 == 0) {
        for (uint32_t i = 0; i < input.n; ++i) output.put(i, input.get(i));
        output.n = input.n;
    }
}

__device__ __forceinline__ uint32_t maxAxis(V3 p) { return (p.x > p.y) ? ((p.x > p.z) ? 0 : 2) : ((p.y > p.z) ? 1 : 2); }

__device__ inline void boxClipPlanes(V3 radius, V3 normal, V3* pts, V3* nrm) {  // 225-256
    V3 p = vabs(normal);
    uint32_t me = maxAxis(p), a0 = (me + 1) % 3, a1 = (me + 2) % 3;
    V3 n0, n1, n2, n3;
    n0.set(a0, 1.f); n1.set(a1, 1.f); n2.set(a0, -1.f); n3.set(a1, -1.f);
    nrm[0] = n0; pts[0] = -radius;
    nrm[1] = n1; pts[1] = -radius;
    nrm[2] = n2; pts[2] = radius;
    nrm[3] = n3; pts[3] = radius;
}

__device__ inline void boxIncidentFace(V3 radius, V3 normal, V3* quad) {  // 259-293
    V3 p = vabs(normal);
    uint32_t me = maxAxis(p), a0 = (me + 1) % 3, a1 = (me + 2) % 3;
    float s = normal.get(me) < 0.f ? 1.f : -1.f;
    float d = radius.get(me) * s;
    float mn0 = -radius.get(a0), mn1 = -radius.get(a1), mx0 = radius.get(a0), mx1 = radius.get(a1);
    float c0[4] = {mn0, mx0, mx0, mn0}, c1[4] = {mn1, mn1, mx1, mx1};
#pragma unroll
    for (int i = 0; i < 4; ++i) { V3 v; v.set(me, d); v.set(a0, c0[i]); v.set(a1, c1[i]); quad[i] = v; }
}

__device__ __forceinline__ P4 boxReferencePlane(V3 mn, V3 mx, V3 normal) {  // 295-303
    V3 point((normal.x < 0.f) ? mn.x : mx.x, (normal.y < 0.f) ? mn.y : mx.y, (normal.z < 0.f) ? mn.z : mx.z);
    return makePlane(point, normal);
}

__device__ inline void boxIncidentEdge(V3 r, V3 normal, V3& oa, V3& ob) {  // 305-337
    V3 p = vabs(normal);
    oa = V3(r.x, r.y, r.z);
    if (p.x > p.y) { if (p.y > p.z) ob = V3(r.x, r.y, -r.z); else ob = V3(r.x, -r.y, r.z); }
    else { if (p.x > p.z) ob = V3(r.x, r.y, -r.z); else ob = V3(-r.x, r.y, r.z); }
    V3 s(normal.x < 0.f ? -1.f : 1.f, normal.y < 0.f ? -1.f : 1.f, normal.z < 0.f ? -1.f : 1.f);
    oa = oa * s; ob = ob * s;
}

template <class Poly>
__device__ inline bool clipAndBuild(Poly& poly, Poly& clipped, const P4* planes, uint32_t numPlanes, P4 ref, Manifold& out) {  // 339-369
    clipPolygon(poly, planes, numPlanes, clipped);
    if (clipped.n > 0) {
        V3 rn(ref.x, ref.y, ref.z);
        for (uint32_t i = 0; i < clipped.n; ++i) {
            ClipVert c = clipped.get(i);
            if (c.depth < 0.f) { clipped.put(i, clipped.get(clipped.n - 1)); --clipped.n; --i; }
            else { c.v = c.v + rn * c.depth; clipped.put(i, c); }
        }
        if (clipped.n > 0) { reduceManifold(clipped, clipped.n, out.n, out); return true; }
    }
    return false;
}

// The same clip + reduction for the LDS polygon of k_narrow_clip, arranged for the machine: every pass first reads the WHOLE polygon into registers
// (<= 8 vertices, the loads issued back to back: one LDS latency per pass instead of one per vertex — the compiler cannot overlap them itself, reads
// and writes of the ping-pong planes may alias for all it knows), then walks it with static indices and writes the surviving vertices back IN PLACE
// (everything a pass needs is already in registers, so one [vertex][lane] plane per lane suffices: 32 KiB per workgroup instead of 64).  The
// arithmetic per vertex, and the order of the vertices, are exactly clipPolygon's / clipAndBuild's / reduceManifold's above.
__device__ inline void clipPolygonLds(LdsPoly& poly, const P4* planes, uint32_t numPlanes) {
    for (uint32_t ci = 0; ci < numPlanes; ++ci) {
        const uint32_t n = poly.n;
        if (n == 0) break;
        const P4 pl = planes[ci];
        ClipVert v[kLdsPolyVerts];
#pragma unroll
        for (uint32_t i = 0; i < kLdsPolyVerts; ++i) if (i < n) v[i] = poly.get(i);
        ClipVert start = poly.get(n - 1);
        uint32_t on = 0;
#pragma unroll
        for (uint32_t i = 0; i < kLdsPolyVerts; ++i) {
            if (i < n) {
                const ClipVert end = v[i];
                const float sd = planeDist(start.v, pl), ed = planeDist(end.v, pl);
                const bool sIn = sd > 0.f, eIn = ed > 0.f;
                if (sIn && eIn) poly.put(on++, end);
                else if (sIn) poly.put(on++, clipEdge(start, end, sd, ed));
                else if (!sIn && eIn) { poly.put(on++, clipEdge(start, end, sd, ed)); poly.put(on++, end); }
                start = end;
            }
        }
        poly.n = on;
    }
}
__device__ inline void reduceManifoldLds(const LdsPoly& poly, uint32_t n, V3 normal, Manifold& out) {
    ClipVert v[kLdsPolyVerts];
#pragma unroll
    for (uint32_t i = 0; i < kLdsPolyVerts; ++i) if (i < n) v[i] = poly.get(i);
    if (n > 4) {
        const V3 searchDir = tangentOf(normal);
        float best = dot(searchDir, v[0].v);
        uint32_t ri = 0;
#pragma unroll
        for (uint32_t i = 1; i < kLdsPolyVerts; ++i) if (i < n) { float dd = dot(searchDir, v[i].v); if (dd > best) { ri = i; best = dd; } }
        { ClipVert c = poly.get(ri); setc(out, 0, c.v, c.depth); }
        best = 0.f; ri = 0;
#pragma unroll
        for (uint32_t i = 0; i < kLdsPolyVerts; ++i) if (i < n) { float sq = sqlen(v[i].v - out.p[0]); if (sq > best) { ri = i; best = sq; } }
        { ClipVert c = poly.get(ri); setc(out, 1, c.v, c.depth); }
        float bestArea = 0.f; ri = 0;
#pragma unroll
        for (uint32_t i = 0; i < kLdsPolyVerts; ++i) if (i < n) {
            V3 vi = v[i].v;
            V3 qa = out.p[0] - vi, qb = out.p[1] - vi;
            float area = 0.5f * dot(cross(qa, qb), normal);
            if (area > bestArea) { ri = i; bestArea = area; }
        }
        { ClipVert c = poly.get(ri); setc(out, 2, c.v, c.depth); }
        bestArea = 0.f; ri = 0;
#pragma unroll
        for (uint32_t i = 0; i < kLdsPolyVerts; ++i) if (i < n) {
            V3 vi = v[i].v;
            V3 qa = out.p[0] - vi, qb = out.p[1] - vi, qc = out.p[2] - vi;
            float a1 = 0.5f * dot(cross(qa, qb), normal);
            float a2 = 0.5f * dot(cross(qb, qc), normal);
            float a3 = 0.5f * dot(cross(qc, qa), normal);
            float area = fmaxr(fmaxr(a1, a2), a3);
            if (area > bestArea) { ri = i; bestArea = area; }
        }
        { ClipVert c = poly.get(ri); setc(out, 3, c.v, c.depth); }
        out.count = 4;
    } else {
        out.count = n;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) if (i < n) setc(out, i, v[i].v, v[i].depth);
    }
}
__device__ inline bool clipAndBuildLds(LdsPoly& poly, const P4* planes, uint32_t numPlanes, P4 ref, Manifold& out) {
    clipPolygonLds(poly, planes, numPlanes);
    if (poly.n > 0) {
        V3 rn(ref.x, ref.y, ref.z);
        for (uint32_t i = 0; i < poly.n; ++i) {       // (swap-and-pop: the order of the survivors is part of the result)
            ClipVert c = poly.get(i);
            if (c.depth < 0.f) { poly.put(i, poly.get(poly.n - 1)); --poly.n; --i; }
            else { c.v = c.v + rn * c.depth; poly.put(i, c); }
        }
        if (poly.n > 0) { reduceManifoldLds(poly, poly.n, out.n, out); return true; }
    }
    return false;
}

__device__ __forceinline__ V3 closestOnSegment(V3 q, V3 la, V3 lb) {  // bounding_volumes.h:365-371
    V3 ab = lb - la;
    float t = dot(q - la, ab) / sqlen(ab);
    t = clampr(t, 0.f, 1.f);
    return la + t * ab;
}
__device__ __forceinline__ V3 closestOnAABB(V3 q, V3 mn, V3 mx) {  // bounding_volumes.h:373-384
    V3 r;
    float v = q.x; if (v < mn.x) v = mn.x; if (v > mx.x) v = mx.x; r.x = v;
    v = q.y; if (v < mn.y) v = mn.y; if (v > mx.y) v = mx.y; r.y = v;
    v = q.z; if (v < mn.z) v = mn.z; if (v > mx.z) v = mx.z; r.z = v;
    return r;
}
__device__ inline float closestSegmentSegment(V3 p1, V3 q1, V3 p2, V3 q2, V3& c1, V3& c2) {  // bounding_volumes.cpp:1251-1315
    float s, t;
    V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    if (a <= kEps && e <= kEps) { c1 = p1; c2 = p2; return dot(c1 - c2, c1 - c2); }
    if (a <= kEps) { s = 0.f; t = f / e; t = clampr(t, 0.f, 1.f); }
    else {
        float c = dot(d1, r);
        if (e <= kEps) { t = 0.f; s = clampr(-c / a, 0.f, 1.f); }
        else {
            float b = dot(d1, d2);
            float denom = a * e - b * b;
            if (denom != 0.f) s = clampr((b * f - c * e) / denom, 0.f, 1.f); else s = 0.f;
            t = (b * s + f) / e;
            if (t < 0.f) { t = 0.f; s = clampr(-c / a, 0.f, 1.f); }
            else if (t > 1.f) { t = 1.f; s = clampr((b - c) / a, 0.f, 1.f); }
        }
    }
    c1 = p1 + d1 * s;
    c2 = p2 + d2 * t;
    return sqlen(c1 - c2);
}

// ---- sphere family (374-521)
__device__ inline bool sphereSphere(V3 c1, float r1, V3 c2, float r2, Manifold& out) {
    V3 n = c2 - c1;
    float rs = r2 + r1;
    float sq = sqlen(n);
    if (sq <= rs * rs) {
        float dist;
        if (sq == 0.f) { dist = 0.f; out.n = V3(0.f, 1.f, 0.f); }
        else { dist = sqrtf(sq); out.n = n / dist; }
        out.count = 1;
        out.d[0] = rs - dist;
        out.p[0] = 0.5f * (c1 + r1 * out.n + c2 - r2 * out.n);
        return true;
    }
    return false;
}
__device__ inline bool sphereCylinder(V3 sc, float sr, V3 ca, V3 cb, float cr, Manifold& out) {  // 409-452
    V3 ab = cb - ca;
    float t = dot(sc - ca, ab) / sqlen(ab);
    if (t >= 0.f && t <= 1.f) return sphereSphere(sc, sr, lerp(ca, cb, t), cr, out);
    V3 p = (t <= 0.f) ? ca : cb;
    V3 up = (t <= 0.f) ? -ab : ab;
    V3 proj = normalize(cross(cross(up, sc - p), up));
    V3 endA = p + proj * cr, endB = p - proj * cr;
    V3 closest = closestOnSegment(sc, endA, endB);
    V3 normal = closest - sc;
    float sq = sqlen(normal);
    if (sq <= sr * sr) {
        float dist;
        if (sq == 0.f) { dist = 0.f; out.n = -normalize(up); }
        else { dist = sqrtf(sq); out.n = normal / dist; }
        out.count = 1;
        out.d[0] = sr - dist;
        out.p[0] = closest + 0.5f * out.d[0] * normal;
        return true;
    }
    return false;
}
__device__ inline bool sphereAABB(V3 sc, float sr, V3 mn, V3 mx, Manifold& out) {  // 454-481
    V3 p = closestOnAABB(sc, mn, mx);
    V3 n = p - sc;
    float sq = sqlen(n);
    if (sq <= sr * sr) {
        float dist = 0.f;
        if (sq > 0.f) { dist = sqrtf(sq); n = n / dist; }
        else n = V3(0.f, 1.f, 0.f);
        out.count = 1;
        out.n = n;
        out.d[0] = sr - dist;
        out.p[0] = 0.5f * (p + sc + n * sr);
        return true;
    }
    return false;
}
__device__ inline bool sphereOBB(V3 sc, float sr, Q4 orot, V3 oc, V3 orad, Manifold& out) {  // 483-497
    V3 mn = oc - orad, mx = oc + orad;
    V3 sl = rotate(conj(orot), sc - oc) + oc;
    if (sphereAABB(sl, sr, mn, mx, out)) {
        out.n = rotate(orot, out.n);
        out.p[0] = rotate(orot, out.p[0] - oc) + oc;
        return true;
    }
    return false;
}

// ---- capsule vs capsule / cylinder (523-703)
__device__ inline bool capsuleVsSegmentShape(const Shape& a, const Shape& b, bool bIsCylinder, Manifold& out) {
    V3 aDir = a.b - a.a;
    V3 bDir = normalize(b.b - b.a);
    float aLen = len(aDir);
    aDir = aDir * (1.f / aLen);
    float parallel = dot(aDir, bDir);
    V3 endA, endB;   // sphere centres for the end / skew fallbacks
    if (fabsf(parallel) > 0.99f) {
        V3 pAa = a.a, pAb = a.b, pBa = b.a, pBb = b.b;
        if (parallel < 0.f) { V3 t = pBa; pBa = pBb; pBb = t; }
        V3 ref = a.a;
        float a0 = 0.f, a1 = aLen;
        float b0 = dot(aDir, pBa - ref), b1 = dot(aDir, pBb - ref);
        float left = fmaxr(a0, b0), right = fminr(a1, b1);
        if (!(right < left)) {
            V3 cA0 = ref + left * aDir, cA1 = ref + right * aDir;
            V3 cB0 = closestOnSegment(cA0, pBa, pBb);
            V3 cB1 = cB0 + (right - left) * aDir;
            V3 normal = cB0 - cA0;
            float d = len(normal);
            if (d < kEps) { d = 0.f; normal = V3(0.f, 1.f, 0.f); }
            else normal = normal / d;
            float pen = (a.radius + b.radius) - d;
            if (pen < 0.f) return false;
            out.n = normal;
            out.count = 2;
            out.d[0] = pen; out.p[0] = (cA0 + cB0) * 0.5f;
            out.d[1] = pen; out.p[1] = (cA1 + cB1) * 0.5f;
            return true;
        }
        if (a0 > b1) { endA = pAa; endB = pBb; } else { endA = pAb; endB = pBa; }
    } else {
        closestSegmentSegment(a.a, a.b, b.a, b.b, endA, endB);
    }
    if (bIsCylinder) return sphereCylinder(endA, a.radius, b.a, b.b, b.radius, out);
    return sphereSphere(endA, a.radius, endB, b.radius, out);
}

// ---- boxes
__device__ inline bool aabbAABB(V3 amn, V3 amx, V3 bmn, V3 bmx, Manifold& out) {  // 1074-1140
    V3 cA = (amn + amx) * 0.5f, cB = (bmn + bmx) * 0.5f;
    V3 rA = (amx - amn) * 0.5f, rB = (bmx - bmn) * 0.5f;
    V3 d = cB - cA;
    V3 p = (rB + rA) - vabs(d);
    if (p.x < 0.f || p.y < 0.f || p.z < 0.f) return false;
    uint32_t me = (p.x < p.y) ? ((p.x < p.z) ? 0 : 2) : ((p.y < p.z) ? 1 : 2);
    float s = d.get(me) < 0.f ? -1.f : 1.f;
    float pen = p.get(me) * s;
    V3 normal; normal.set(me, s);
    out.n = normal;
    out.count = 4;
    uint32_t a0 = (me + 1) % 3, a1 = (me + 2) % 3;
    float mn0 = fmaxr(amn.get(a0), bmn.get(a0)), mn1 = fmaxr(amn.get(a1), bmn.get(a1));
    float mx0 = fminr(amx.get(a0), bmx.get(a0)), mx1 = fminr(amx.get(a1), bmx.get(a1));
    float depth = cA.get(me) + rA.get(me) - pen * 0.5f;
    float c0[4] = {mn0, mn0, mx0, mx0}, c1[4] = {mn1, mx1, mn1, mx1};
    for (int i = 0; i < 4; ++i) { V3 pt; pt.set(a0, c0[i]); pt.set(a1, c1[i]); pt.set(me, depth); out.p[i] = pt; out.d[i] = pen; }
    return true;
}

__device__ __forceinline__ V3 obbSupport(Q4 rot, V3 center, V3 radius, V3 dir) {  // collision_gjk.h:59-75
    dir = rotate(conj(rot), dir);
    V3 r(dir.x < 0.f ? -radius.x : radius.x, dir.y < 0.f ? -radius.y : radius.y, dir.z < 0.f ? -radius.z : radius.z);
    return center + rotate(rot, r);
}

// OBB vs OBB (1179-1527), in two halves so a kernel can run the cheap 15-axis SAT for every pair and the expensive
// contact generation (clipping) only for the pairs that overlap, re-packed into dense waves.
struct ObbSat { V3 normal; bool faceHit, bFace; };
__device__ inline bool obbSat(Q4 arot, V3 acen, V3 arad, Q4 brot, V3 bcen, V3 brad, ObbSat& res) {
    V3 ax = rotate(arot, V3(1.f, 0.f, 0.f)), ay = rotate(arot, V3(0.f, 1.f, 0.f)), az = rotate(arot, V3(0.f, 0.f, 1.f));
    V3 bx = rotate(brot, V3(1.f, 0.f, 0.f)), by = rotate(brot, V3(0.f, 1.f, 0.f)), bz = rotate(brot, V3(0.f, 0.f, 1.f));
    M3 r;
    r.m00 = dot(ax, bx); r.m10 = dot(ay, bx); r.m20 = dot(az, bx);
    r.m01 = dot(ax, by); r.m11 = dot(ay, by); r.m21 = dot(az, by);
    r.m02 = dot(ax, bz); r.m12 = dot(ay, bz); r.m22 = dot(az, bz);
    V3 tw = bcen - acen;
    V3 t = rotate(conj(arot), tw);
    M3 q;  // |r| + eps
    q.m00 = fabsf(r.m00) + kEps; q.m01 = fabsf(r.m01) + kEps; q.m02 = fabsf(r.m02) + kEps;
    q.m10 = fabsf(r.m10) + kEps; q.m11 = fabsf(r.m11) + kEps; q.m12 = fabsf(r.m12) + kEps;
    q.m20 = fabsf(r.m20) + kEps; q.m21 = fabsf(r.m21) + kEps; q.m22 = fabsf(r.m22) + kEps;
    bool parallel = q.m00 >= 0.99f || q.m01 >= 0.99f || q.m02 >= 0.99f || q.m10 >= 0.99f || q.m11 >= 0.99f || q.m12 >= 0.99f ||
                    q.m20 >= 0.99f || q.m21 >= 0.99f || q.m22 >= 0.99f;
    float ra, rb;
    float minPen = FLT_MAX;
    V3 normal;
    bool bFace = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ra = arad.get(i);
        rb = dot(q.r(i), brad);
        float d = t.get(i);
        float pen = ra + rb - fabsf(d);
        if (pen < 0.f) return false;
        if (pen < minPen) { minPen = pen; normal = V3(); normal.set(i, 1.f); }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ra = dot(q.c(i), arad);
        rb = brad.get(i);
        float d = dot(r.c(i), t);
        float pen = ra + rb - fabsf(d);
        if (pen < 0.f) return false;
        if (pen < minPen) { minPen = pen; normal = V3(); normal.set(i, 1.f); bFace = true; }
    }
    bool edgeHit = false;
    V3 edgeNormal;
    if (!parallel) {
        float pen, l; V3 n;
#define MI_EDGE(RA, RB, DIST, NX, NY, NZ) \
        ra = RA; rb = RB; pen = ra + rb - fabsf(DIST); \
        if (pen < 0.f) return false; \
        n = V3(NX, NY, NZ); l = 1.f / len(n); pen *= l; \
        if (pen < minPen) { minPen = pen; edgeNormal = n * l; edgeHit = true; }
        MI_EDGE(arad.y * q.m20 + arad.z * q.m10, brad.y * q.m02 + brad.z * q.m01, t.z * r.m10 - t.y * r.m20, 0.f, -r.m20, r.m10)
        MI_EDGE(arad.y * q.m21 + arad.z * q.m11, brad.x * q.m02 + brad.z * q.m00, t.z * r.m11 - t.y * r.m21, 0.f, -r.m21, r.m11)
        MI_EDGE(arad.y * q.m22 + arad.z * q.m12, brad.x * q.m01 + brad.y * q.m00, t.z * r.m12 - t.y * r.m22, 0.f, -r.m22, r.m12)
        MI_EDGE(arad.x * q.m20 + arad.z * q.m00, brad.y * q.m12 + brad.z * q.m11, t.x * r.m20 - t.z * r.m00, r.m20, 0.f, -r.m00)
        MI_EDGE(arad.x * q.m21 + arad.z * q.m01, brad.x * q.m12 + brad.z * q.m10, t.x * r.m21 - t.z * r.m01, r.m21, 0.f, -r.m01)
        MI_EDGE(arad.x * q.m22 + arad.z * q.m02, brad.x * q.m11 + brad.y * q.m10, t.x * r.m22 - t.z * r.m02, r.m22, 0.f, -r.m02)
        MI_EDGE(arad.x * q.m10 + arad.y * q.m00, brad.y * q.m22 + brad.z * q.m21, t.y * r.m00 - t.x * r.m10, -r.m10, r.m00, 0.f)
        MI_EDGE(arad.x * q.m11 + arad.y * q.m01, brad.x * q.m22 + brad.z * q.m20, t.y * r.m01 - t.x * r.m11, -r.m11, r.m01, 0.f)
        MI_EDGE(arad.x * q.m12 + arad.y * q.m02, brad.x * q.m21 + brad.y * q.m20, t.y * r.m02 - t.x * r.m12, -r.m12, r.m02, 0.f)
#undef MI_EDGE
    }
    bool faceHit = !edgeHit;
    if (faceHit) { if (bFace) normal = mul(r, normal); }
    else normal = edgeNormal;
    normal = rotate(arot, normal);
    if (dot(normal, tw) < 0.f) normal = -normal;
    res.normal = normal; res.faceHit = faceHit; res.bFace = bFace;
    return true;
}
template <class Poly>
__device__ inline bool obbContacts(Q4 arot, V3 acen, V3 arad, Q4 brot, V3 bcen, V3 brad, const ObbSat& res, Poly& poly, Poly& clipped, Manifold& out) {
    const V3 normal = res.normal;
    const bool faceHit = res.faceHit, bFace = res.bFace;
    out.n = normal;
    if (faceHit) {
        V3 cp[4], cn[4], quad[4];
        P4 plane;
        if (!bFace) {
            boxClipPlanes(arad, rotate(conj(arot), normal), cp, cn);
            boxIncidentFace(brad, rotate(conj(brot), normal), quad);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                cp[i] = rotate(arot, cp[i]) + acen;
                cn[i] = rotate(arot, cn[i]);
                quad[i] = rotate(brot, quad[i]) + bcen;
            }
            plane = makePlane(obbSupport(arot, acen, arad, normal), normal);
        } else {
            boxClipPlanes(brad, rotate(conj(brot), -normal), cp, cn);
            boxIncidentFace(arad, rotate(conj(arot), -normal), quad);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                cp[i] = rotate(brot, cp[i]) + bcen;
                cn[i] = rotate(brot, cn[i]);
                quad[i] = rotate(arot, quad[i]) + acen;
            }
            plane = makePlane(obbSupport(brot, bcen, brad, -normal), -normal);
        }
        P4 planes[4];
        poly.n = 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            planes[i] = makePlane(cp[i], cn[i]);
            ClipVert c; c.v = quad[i]; c.depth = -planeDist(quad[i], plane);
            poly.put(i, c);
        }
        if (!clipAndBuild(poly, clipped, planes, 4, plane, out)) return false;
    } else {
        V3 a0, a1, b0, b1;
        boxIncidentEdge(arad, rotate(conj(arot), normal), a0, a1);
        boxIncidentEdge(brad, rotate(conj(brot), -normal), b0, b1);
        a0 = rotate(arot, a0) + acen; a1 = rotate(arot, a1) + acen;
        b0 = rotate(brot, b0) + bcen; b1 = rotate(brot, b1) + bcen;
        V3 pa, pb;
        float sq = closestSegmentSegment(a0, a1, b0, b1, pa, pb);
        out.count = 1;
        out.d[0] = sqrtf(sq);
        out.p[0] = (pa + pb) * 0.5f;
    }
    return true;
}
// obbContacts for k_narrow_clip: the face case clips in place in one LDS polygon (clipAndBuildLds); everything else is obbContacts itself
__device__ inline bool obbContactsLds(Q4 arot, V3 acen, V3 arad, Q4 brot, V3 bcen, V3 brad, const ObbSat& res, LdsPoly& poly, Manifold& out) {
    if (!res.faceHit) { LdsPoly none{poly.p, 0u}; return obbContacts(arot, acen, arad, brot, bcen, brad, res, poly, none, out); }   // (edge case: no polygon involved)
    const V3 normal = res.normal;
    const bool bFace = res.bFace;
    out.n = normal;
    V3 cp[4], cn[4], quad[4];
    P4 plane;
    if (!bFace) {
        boxClipPlanes(arad, rotate(conj(arot), normal), cp, cn);
        boxIncidentFace(brad, rotate(conj(brot), normal), quad);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cp[i] = rotate(arot, cp[i]) + acen;
            cn[i] = rotate(arot, cn[i]);
            quad[i] = rotate(brot, quad[i]) + bcen;
        }
        plane = makePlane(obbSupport(arot, acen, arad, normal), normal);
    } else {
        boxClipPlanes(brad, rotate(conj(brot), -normal), cp, cn);
        boxIncidentFace(arad, rotate(conj(arot), -normal), quad);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cp[i] = rotate(brot, cp[i]) + bcen;
            cn[i] = rotate(brot, cn[i]);
            quad[i] = rotate(arot, quad[i]) + acen;
        }
        plane = makePlane(obbSupport(brot, bcen, brad, -normal), -normal);
    }
    P4 planes[4];
    poly.n = 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        planes[i] = makePlane(cp[i], cn[i]);
        ClipVert c; c.v = quad[i]; c.depth = -planeDist(quad[i], plane);
        poly.put(i, c);
    }
    return clipAndBuildLds(poly, planes, 4, plane, out);
}
template <class Poly>
__device__ inline bool obbOBB(Q4 arot, V3 acen, V3 arad, Q4 brot, V3 bcen, V3 brad, Poly& poly, Poly& clipped, Manifold& out) {
    ObbSat res;
    if (!obbSat(arot, acen, arad, brot, bcen, brad, res)) return false;
    return obbContacts(arot, acen, arad, brot, bcen, brad, res, poly, clipped, out);
}

}  // namespace mi
