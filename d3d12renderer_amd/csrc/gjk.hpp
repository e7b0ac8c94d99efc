// gjk.hpp — GJK + EPA on the device, one pair per lane (src/physics/collision_gjk.{h,cpp},
// src/physics/collision_epa.{h,cpp}) and the tests built on them (capsule/cylinder vs box with face
// clipping, cylinder vs cylinder, every *-hull pair; src/physics/collision_narrow.cpp:496-1071,1150-1584).
//
// The reference's EPA polytope is a 73 KB stack object (3 x 1024-entry arrays); with <= 20 iterations at
// most 24 points, 274 triangles and 276 edges can ever exist, so the per-lane state is 24/288/288 entries
// (~12 KB of scratch) without changing behaviour.  GJK pairs run in their own kernel (k_narrow_gjk) so the
// primitive buckets keep a scratch-free kernel.
#pragma once
#include "kernels.hpp"

namespace mi {

struct SupPt { V3 a, b, m; };   // support point on A, on B, Minkowski difference
struct Simplex { SupPt a, b, c, d; uint32_t n; };

__device__ inline V3 supportOf(const Shape& s, const HullSet& hs, V3 dir) {  // collision_gjk.h:6-100
    switch (s.type) {
        case T_SPHERE: return normalize(dir) * s.radius + s.a;
        case T_CAPSULE: {
            float da = dot(dir, s.a), db = dot(dir, s.b);
            V3 far = da > db ? s.a : s.b;
            return normalize(dir) * s.radius + far;
        }
        case T_CYLINDER: {
            float da = dot(dir, s.a), db = dot(dir, s.b);
            V3 far = da > db ? s.a : s.b;
            V3 n = s.a - s.b;
            V3 pd = noz(cross(cross(n, dir), n));
            return far + pd * s.radius;
        }
        case T_AABB:
            return V3((dir.x < 0.f) ? s.a.x : s.b.x, (dir.y < 0.f) ? s.a.y : s.b.y, (dir.z < 0.f) ? s.a.z : s.b.z);
        case T_OBB: {
            dir = rotate(conj(s.rot), dir);
            V3 r(dir.x < 0.f ? -s.b.x : s.b.x, dir.y < 0.f ? -s.b.y : s.b.y, dir.z < 0.f ? -s.b.z : s.b.z);
            return s.a + rotate(s.rot, r);
        }
        default: {
            dir = rotate(conj(s.rot), dir);
            V3 best;
            float maxD = -FLT_MAX;
            uint32_t first = hs.ranges[2 * s.hull], count = hs.ranges[2 * s.hull + 1];
            for (uint32_t i = 0; i < count; ++i) {
                V3 v = xyz(hs.verts[first + i]);
                float d = dot(dir, v);
                if (d > maxD) { maxD = d; best = v; }
            }
            return s.a + rotate(s.rot, best);
        }
    }
}

__device__ inline SupPt supportPair(const Shape& A, const Shape& B, const HullSet& hs, V3 dir) {
    SupPt p;
    p.a = supportOf(A, hs, dir);
    p.b = supportOf(B, hs, -dir);
    p.m = p.a - p.b;
    return p;
}
__device__ __forceinline__ V3 crossABA(V3 a, V3 b) { return cross(cross(a, b), a); }

enum : int { GJK_STOP = 0, GJK_CONT = 1, GJK_ERR = 2 };

// collision_gjk.cpp:6-212; the goto labels become an entry index into a fall-through switch.
__device__ inline int updateSimplex(Simplex& s, const SupPt& a, V3& dir) {
    if (s.n == 2) {
        V3 ao = -a.m, ab = s.b.m - a.m, ac = s.c.m - a.m;
        V3 abc = cross(ab, ac);
        V3 abp = cross(ab, abc);
        if (dot(ao, abp) > 0.f) { s.c = a; dir = crossABA(ab, ao); return GJK_CONT; }
        V3 acp = cross(abc, ac);
        if (dot(ao, acp) > 0.f) { s.b = a; dir = crossABA(ac, ao); return GJK_CONT; }
        if (dot(ao, abc) >= 0.f) { s.d = s.b; s.b = a; s.n = 3; dir = abc; return GJK_CONT; }
        if (dot(ao, -abc) >= 0.f) { s.d = s.c; s.c = s.b; s.b = a; s.n = 3; dir = -abc; return GJK_CONT; }
        return GJK_ERR;
    }
    if (s.n == 3) {
        V3 ao = -a.m, ab = s.b.m - a.m, ac = s.c.m - a.m, ad = s.d.m - a.m;
        V3 bcd = cross(s.c.m - s.b.m, s.d.m - s.b.m);
        if (dot(bcd, dir) > 0.00001f || dot(bcd, s.b.m) < -0.00001f) return GJK_ERR;
        V3 abc = cross(ac, ab), abd = cross(ab, ad), adc = cross(ad, ac);
        int flags = 0;
        flags |= (dot(abc, ao) > 0.f) ? 1 : 0;
        flags |= (dot(abd, ao) > 0.f) ? 2 : 0;
        flags |= (dot(adc, ao) > 0.f) ? 4 : 0;
        if (flags == 7) return GJK_ERR;
        if (flags == 0) return GJK_STOP;
        int label = 0;   // 1 overABC1, 2 overABC2, 3 overABD1, 4 overABD2, 5 overADC1, 6 overADC2
        if (flags == 1) label = 1;
        else if (flags == 2) label = 3;
        else if (flags == 4) label = 5;
        else if (flags == 3) label = (dot(cross(abc, ab), ao) > 0.f) ? 3 : 2;
        else if (flags == 6) label = (dot(cross(abd, ad), ao) > 0.f) ? 5 : 4;
        else if (flags == 5) label = (dot(cross(adc, ac), ao) > 0.f) ? 1 : 6;
        switch (label) {
            case 1:
                if (dot(cross(abc, ab), ao) > 0.f) { s.c = a; s.n = 2; dir = crossABA(ab, ao); return GJK_CONT; }
            case 2:
                if (dot(cross(ac, abc), ao) > 0.f) { s.b = a; s.n = 2; dir = crossABA(ac, ao); return GJK_CONT; }
                s.d = a; dir = abc; return GJK_CONT;
            case 3:
                if (dot(cross(abd, ad), ao) > 0.f) { s.b = s.d; s.c = a; s.n = 2; dir = crossABA(ad, ao); return GJK_CONT; }
            case 4:
                if (dot(cross(ab, abd), ao) > 0.f) { s.c = a; s.n = 2; dir = crossABA(ab, ao); return GJK_CONT; }
                s.c = a; dir = abd; return GJK_CONT;
            case 5:
                if (dot(cross(adc, ac), ao) > 0.f) { s.b = a; s.n = 2; dir = crossABA(ac, ao); return GJK_CONT; }
            case 6:
                if (dot(cross(ad, adc), ao) > 0.f) { s.b = a; s.c = s.d; s.n = 2; dir = crossABA(ad, ao); return GJK_CONT; }
                s.b = a; dir = adc; return GJK_CONT;
        }
        return GJK_ERR;
    }
    return GJK_ERR;
}

// collision_gjk.h:182-238 (+ a 64-iteration guard: a lane must terminate; treated like the reference's error path)
__device__ inline bool gjkTest(const Shape& A, const Shape& B, const HullSet& hs, Simplex& sx) {
    V3 dir(1.f, 0.1f, -0.2f);
    sx.n = 0;
    sx.c = supportPair(A, B, hs, dir);
    if (dot(sx.c.m, dir) < 0.f) return false;
    dir = -sx.c.m;
    sx.b = supportPair(A, B, hs, dir);
    if (dot(sx.b.m, dir) < 0.f) return false;
    dir = crossABA(sx.c.m - sx.b.m, -sx.b.m);
    sx.n = 2;
    for (int guard = 0; guard < 64; ++guard) {
        if (sqlen(dir) < 0.0001f) return false;
        SupPt a = supportPair(A, B, hs, dir);
        if (dot(a.m, dir) < 0.f) return false;
        int r = updateSimplex(sx, a, dir);
        if (r == GJK_STOP) { sx.a = a; sx.n = 4; return true; }
        if (r == GJK_ERR) return false;
    }
    return false;
}

// ---- boolean overlap tests for triggers / force fields: overlapCheck, src/physics/collision_narrow.cpp:1586-1689, dispatching to
// src/physics/bounding_volumes.h:301-363 and bounding_volumes.cpp:704-835, 1079-1244.  a.type <= b.type.
__device__ inline bool sphereSphereB(V3 ca, float ra, V3 cb, float rb) {
    V3 d = ca - cb;
    float dist2 = dot(d, d), rs = ra + rb;
    return dist2 <= rs * rs;
}
__device__ inline bool sphereCylinderB(V3 sc, float sr, V3 ca, V3 cb, float cr) {   // compares a squared distance with the radius, as written in the reference
    V3 ab = cb - ca;
    float t = dot(sc - ca, ab) / sqlen(ab);
    if (t >= 0.f && t <= 1.f) return sphereSphereB(sc, sr, lerp(ca, cb, t), cr);
    V3 p = (t <= 0.f) ? ca : cb;
    V3 up = (t <= 0.f) ? -ab : ab;
    V3 proj = normalize(cross(cross(up, sc - p), up));
    V3 endA = p + proj * cr, endB = p - proj * cr;
    V3 closest = closestOnSegment(sc, endA, endB);
    return sqlen(closest - sc) <= sr;
}
__device__ inline bool sphereAABBB(V3 sc, float sr, V3 mn, V3 mx) {
    V3 n = closestOnAABB(sc, mn, mx) - sc;
    return sqlen(n) <= sr * sr;
}
__device__ inline bool obbObbB(Q4 arot, V3 acen, V3 arad, Q4 brot, V3 bcen, V3 brad) {   // bounding_volumes.cpp:1079-1199: all 15 axes, no parallel shortcut
    V3 ax = rotate(arot, V3(1.f, 0.f, 0.f)), ay = rotate(arot, V3(0.f, 1.f, 0.f)), az = rotate(arot, V3(0.f, 0.f, 1.f));
    V3 bx = rotate(brot, V3(1.f, 0.f, 0.f)), by = rotate(brot, V3(0.f, 1.f, 0.f)), bz = rotate(brot, V3(0.f, 0.f, 1.f));
    M3 r;
    r.m00 = dot(ax, bx); r.m10 = dot(ay, bx); r.m20 = dot(az, bx);
    r.m01 = dot(ax, by); r.m11 = dot(ay, by); r.m21 = dot(az, by);
    r.m02 = dot(ax, bz); r.m12 = dot(ay, bz); r.m22 = dot(az, bz);
    V3 tw = bcen - acen;
    V3 t = rotate(conj(arot), tw);
    M3 q;
    q.m00 = fabsf(r.m00) + kEps; q.m01 = fabsf(r.m01) + kEps; q.m02 = fabsf(r.m02) + kEps;
    q.m10 = fabsf(r.m10) + kEps; q.m11 = fabsf(r.m11) + kEps; q.m12 = fabsf(r.m12) + kEps;
    q.m20 = fabsf(r.m20) + kEps; q.m21 = fabsf(r.m21) + kEps; q.m22 = fabsf(r.m22) + kEps;
    float ra, rb;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ra = arad.get(i); rb = dot(q.r(i), brad);
        if (ra + rb - fabsf(t.get(i)) < 0.f) return false;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ra = dot(q.c(i), arad); rb = brad.get(i);
        if (ra + rb - fabsf(dot(r.c(i), t)) < 0.f) return false;
    }
#define MI_EDGEB(RA, RB, D) ra = RA; rb = RB; if (ra + rb - fabsf(D) < 0.f) return false;
    MI_EDGEB(arad.y * q.m20 + arad.z * q.m10, brad.y * q.m02 + brad.z * q.m01, t.z * r.m10 - t.y * r.m20)
    MI_EDGEB(arad.y * q.m21 + arad.z * q.m11, brad.x * q.m02 + brad.z * q.m00, t.z * r.m11 - t.y * r.m21)
    MI_EDGEB(arad.y * q.m22 + arad.z * q.m12, brad.x * q.m01 + brad.y * q.m00, t.z * r.m12 - t.y * r.m22)
    MI_EDGEB(arad.x * q.m20 + arad.z * q.m00, brad.y * q.m12 + brad.z * q.m11, t.x * r.m20 - t.z * r.m00)
    MI_EDGEB(arad.x * q.m21 + arad.z * q.m01, brad.x * q.m12 + brad.z * q.m10, t.x * r.m21 - t.z * r.m01)
    MI_EDGEB(arad.x * q.m22 + arad.z * q.m02, brad.x * q.m11 + brad.y * q.m10, t.x * r.m22 - t.z * r.m02)
    MI_EDGEB(arad.x * q.m10 + arad.y * q.m00, brad.y * q.m22 + brad.z * q.m21, t.y * r.m00 - t.x * r.m10)
    MI_EDGEB(arad.x * q.m11 + arad.y * q.m01, brad.x * q.m22 + brad.z * q.m20, t.y * r.m01 - t.x * r.m11)
    MI_EDGEB(arad.x * q.m12 + arad.y * q.m02, brad.x * q.m21 + brad.y * q.m20, t.y * r.m02 - t.x * r.m12)
#undef MI_EDGEB
    return true;
}
__device__ inline bool gjkBool(const Shape& a, const Shape& b, const HullSet& hs) { Simplex sx; return gjkTest(a, b, hs, sx); }
__device__ inline bool segmentVsObbB(const Shape& c, const Shape& o, const HullSet& hs) {   // capsuleVsOBB / cylinderVsOBB: into the box frame, then GJK vs the AABB
    Shape box; box.type = T_AABB; box.a = o.a - o.b; box.b = o.a + o.b; box.radius = 0.f; box.hull = 0;
    Shape r = c;
    r.a = rotate(conj(o.rot), c.a - o.a) + o.a;
    r.b = rotate(conj(o.rot), c.b - o.a) + o.a;
    return gjkBool(r, box, hs);
}
__device__ inline bool overlapCheck(const Shape& a, const Shape& b, const HullSet& hs) {
    switch (a.type) {
        case T_SPHERE:
            switch (b.type) {
                case T_SPHERE: return sphereSphereB(a.a, a.radius, b.a, b.radius);
                case T_CAPSULE: return sphereSphereB(a.a, a.radius, closestOnSegment(a.a, b.a, b.b), b.radius);
                case T_CYLINDER: return sphereCylinderB(a.a, a.radius, b.a, b.b, b.radius);
                case T_AABB: return sphereAABBB(a.a, a.radius, b.a, b.b);
                case T_OBB: return sphereAABBB(rotate(conj(b.rot), a.a - b.a) + b.a, a.radius, b.a - b.b, b.a + b.b);
                default: return gjkBool(a, b, hs);
            }
        case T_CAPSULE:
            switch (b.type) {
                case T_CAPSULE: { V3 c1, c2; closestSegmentSegment(a.a, a.b, b.a, b.b, c1, c2); return sphereSphereB(c1, a.radius, c2, b.radius); }
                case T_CYLINDER: { V3 c1, c2; closestSegmentSegment(a.a, a.b, b.a, b.b, c1, c2); return sphereCylinderB(c1, a.radius, b.a, b.b, b.radius); }
                case T_OBB: return segmentVsObbB(a, b, hs);
                default: return gjkBool(a, b, hs);
            }
        case T_CYLINDER:
            if (b.type == T_OBB) return segmentVsObbB(a, b, hs);
            return gjkBool(a, b, hs);
        case T_AABB:
            switch (b.type) {
                case T_AABB:
                    if (a.b.x < b.a.x || a.a.x > b.b.x) return false;
                    if (a.b.y < b.a.y || a.a.y > b.b.y) return false;
                    if (a.b.z < b.a.z || a.a.z > b.b.z) return false;
                    return true;
                case T_OBB: return obbObbB(Q4(0.f, 0.f, 0.f, 1.f), (a.a + a.b) * 0.5f, (a.b - a.a) * 0.5f, b.rot, b.a, b.b);
                default: return gjkBool(a, b, hs);
            }
        case T_OBB:
            if (b.type == T_OBB) return obbObbB(a.rot, a.a, a.b, b.rot, b.a, b.b);
            return gjkBool(a, b, hs);
        default:
            return gjkBool(a, b, hs);
    }
}

// One lane per (rigid-body collider, trigger / force-field collider) AABB-overlap: the boolean test; hits are appended as
// non_collision_interaction records (body, other object index | type << 28, body collider, other collider).
struct DeviceInteraction { uint32_t body, other, rbCollider, otherCollider; };
__global__ __launch_bounds__(64) void k_overlap(StepScalars* sc, uint32_t cap, const uint64_t* __restrict__ interKeys, const float4* __restrict__ wShape,
                                                const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax, HullSet hs,
                                                DeviceInteraction* __restrict__ out) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= sc->numInterPairs) return;
    uint64_t key = interKeys[p];
    uint32_t bucket = (uint32_t)(key >> 58), a = (uint32_t)((key >> 29) & 0x1FFFFFFFu), b = (uint32_t)(key & 0x1FFFFFFFu);
    uint32_t ta = 0, rem = bucket;
    while (rem >= 6u - ta) { rem -= 6u - ta; ++ta; }
    uint32_t tb = ta + rem;
    Shape sa = loadShape(wShape, a, ta), sb = loadShape(wShape, b, tb);
    if (!overlapCheck(sa, sb, hs)) return;
    uint32_t tagA = __float_as_uint(aabbMin[a].w), tagB = __float_as_uint(aabbMin[b].w);
    uint32_t oa = (tagA >> 8) & 0xFFu, ob = (tagB >> 8) & 0xFFu;
    uint32_t ia = __float_as_uint(aabbMax[a].w), ib = __float_as_uint(aabbMax[b].w);
    DeviceInteraction in;
    if (oa == OBJ_RIGID_BODY) { in.body = ia; in.other = ib | (ob << 28); in.rbCollider = a; in.otherCollider = b; }
    else { in.body = ib; in.other = ia | (oa << 28); in.rbCollider = b; in.otherCollider = a; }
    uint32_t slot = atomicAdd(&sc->numInteractions, 1u);
    if (slot < cap) out[slot] = in;
}
// Localized force fields: the host sorted the interactions into the canonical order (body, other collider, body collider);
// the first lane of every body's segment adds the forces sequentially, like rb.forceAccumulator += ff.force in that order.
__global__ __launch_bounds__(256) void k_apply_fields(uint32_t n, const uint2* __restrict__ sorted /* (body, force-field index) */,
                                                      const float4* __restrict__ localForce, float4* __restrict__ bForceStep) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint2 me = sorted[i];
    if (i > 0 && sorted[i - 1].x == me.x) return;
    V3 f = xyz(bForceStep[me.x]);
    for (uint32_t j = i; j < n && sorted[j].x == me.x; ++j) f = f + xyz(localForce[sorted[j].y]);
    bForceStep[me.x] = f4(f, 0.f);
}

// The same without the host (speculative steps): the interactions are put into the canonical order ON THE DEVICE — their keys
// (body, other collider, body collider) are unique, so the sorted position of an entry is the number of entries with a smaller key
// (a rank sort: n x n comparisons through LDS tiles; n is hundreds to a few thousand, the host path takes over beyond `cap`) — and the
// first entry of every body's run adds that body's force-field forces in order.  Trigger overlaps are read from the sorted list
// after the step.
__device__ __forceinline__ bool interLess(const DeviceInteraction& x, const DeviceInteraction& y) {
    if (x.body != y.body) return x.body < y.body;
    if (x.otherCollider != y.otherCollider) return x.otherCollider < y.otherCollider;
    return x.rbCollider < y.rbCollider;
}
__global__ __launch_bounds__(256) void k_inter_sort(const StepScalars* __restrict__ sc, uint32_t cap, const DeviceInteraction* __restrict__ in, DeviceInteraction* __restrict__ out) {
    __shared__ DeviceInteraction tile[256];
    const uint32_t n = min(sc->numInteractions, cap);
    if (blockIdx.x * 256u >= n) return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    DeviceInteraction me{0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (i < n) me = in[i];
    uint32_t rank = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += 256u) {
        __syncthreads();
        if (j0 + threadIdx.x < n) tile[threadIdx.x] = in[j0 + threadIdx.x];
        __syncthreads();
        const uint32_t m = min(256u, n - j0);
        for (uint32_t j = 0; j < m; ++j) rank += interLess(tile[j], me) ? 1u : 0u;
    }
    if (i < n) out[rank] = me;
}
__global__ __launch_bounds__(256) void k_apply_fields_sorted(const StepScalars* __restrict__ sc, uint32_t cap, const DeviceInteraction* __restrict__ sorted,
                                                             const float4* __restrict__ localForce, float4* __restrict__ bForceStep) {
    const uint32_t n = min(sc->numInteractions, cap);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t body = sorted[i].body;
    if (i > 0 && sorted[i - 1].body == body) return;
    V3 f = xyz(bForceStep[body]); bool any = false;
    for (uint32_t j = i; j < n && sorted[j].body == body; ++j) {
        const uint32_t other = sorted[j].other;
        if ((other >> 28) == OBJ_FORCE_FIELD) { f = f + xyz(localForce[other & 0x0FFFFFFFu]); any = true; }
    }
    if (any) bForceStep[body] = f4(f, 0.f);
}

// ---- ray tests (testPhysicsInteraction, src/physics/physics.cpp:555-629; ray::intersect*, bounding_volumes.cpp:197-398, 677-702).
// Stated deviation: the cylinder test's outT is 0 (not uninitialised) when the origin is inside the infinite cylinder and
// neither cap is hit.
__device__ inline bool rayPlane(V3 o, V3 d, V3 normal, float pd, float& t) {
    float ndotd = dot(d, normal);
    if (fabsf(ndotd) < 1e-6f) return false;
    t = -(dot(o, normal) + pd) / ndotd;
    return true;
}
__device__ inline bool rayDisk(V3 o, V3 d, V3 pos, V3 normal, float radius, float& t) {
    if (!rayPlane(o, d, normal, -dot(normal, pos), t)) return false;
    return len(o + t * d - pos) <= radius;
}
__device__ inline bool raySphere(V3 o, V3 d, V3 center, float radius, float& t) {
    V3 m = o - center;
    float b = dot(m, d), c = dot(m, m) - radius * radius;
    if (c > 0.f && b > 0.f) return false;
    float discr = b * b - c;
    if (discr < 0.f) return false;
    t = -b - sqrtf(discr);
    if (t < 0.f) t = 0.f;
    return true;
}
__device__ inline bool rayCylinder(V3 o, V3 d, V3 pa, V3 pb, float radius, float& t) {
    V3 axis = pb - pa;
    float height = len(axis);
    Q4 q = rotateFromTo(axis, V3(0.f, 1.f, 0.f));
    o = rotate(q, o - pa); d = rotate(q, d);
    const float epsilon = 1e-6f;
    float y = -1.f;
    t = 0.f;
    if (o.x * o.x + o.z * o.z > radius * radius) {
        float a = d.x * d.x + d.z * d.z, b = d.x * o.x + d.z * o.z, c = o.x * o.x + o.z * o.z - radius * radius;
        float delta = b * b - a * c;
        if (delta < epsilon) return false;
        t = (-b - sqrtf(delta)) / a;
        if (t <= epsilon) return false;
        y = o.y + t * d.y;
    }
    if (y > height + epsilon || y < -epsilon) {
        float dist;
        if (d.y < 0.f && rayDisk(o, d, V3(0.f, height, 0.f), V3(0.f, 1.f, 0.f), radius, dist)) t = dist;
        if (d.y > 0.f && rayDisk(o, d, V3(0.f, 0.f, 0.f), V3(0.f, -1.f, 0.f), radius, dist)) t = dist;
        y = o.y + t * d.y;
    }
    return y > -epsilon && y < height + epsilon;
}
__device__ inline bool rayAABB(V3 o, V3 d, V3 mn, V3 mx, float& t) {
    V3 inv(1.f / d.x, 1.f / d.y, 1.f / d.z);
    float tx1 = (mn.x - o.x) * inv.x, tx2 = (mx.x - o.x) * inv.x;
    t = fminr(tx1, tx2);
    float tmax = fmaxr(tx1, tx2);
    float ty1 = (mn.y - o.y) * inv.y, ty2 = (mx.y - o.y) * inv.y;
    t = fmaxr(t, fminr(ty1, ty2)); tmax = fminr(tmax, fmaxr(ty1, ty2));
    float tz1 = (mn.z - o.z) * inv.z, tz2 = (mx.z - o.z) * inv.z;
    t = fmaxr(t, fminr(tz1, tz2)); tmax = fminr(tmax, fmaxr(tz1, tz2));
    return tmax >= t && t > 0.f;
}
__device__ inline bool pointInTriangle(V3 point, V3 a, V3 b, V3 c) {   // math.cpp:1273-1290
    V3 e10 = b - a, e20 = c - a;
    float aa = dot(e10, e10), bb = dot(e10, e20), cc = dot(e20, e20);
    float ac_bb = (aa * cc) - (bb * bb);
    V3 vp = point - a;
    float dd = dot(vp, e10), ee = dot(vp, e20);
    float x = (dd * cc) - (ee * bb), y = (ee * aa) - (dd * bb), z = x + y - ac_bb;
    return ((__float_as_uint(z) & ~(__float_as_uint(x) | __float_as_uint(y))) & 0x80000000u) != 0u;
}
__device__ inline bool rayTriangle(V3 o, V3 d, V3 a, V3 b, V3 c, float& t) {
    V3 normal = noz(cross(b - a, c - a));
    float pd = -dot(normal, a);
    float nDotR = dot(d, normal);
    if (fabsf(nDotR) <= 1e-6f) return false;
    t = -(dot(o, normal) + pd) / nDotR;
    V3 q = o + t * d;
    return t >= 0.f && pointInTriangle(q, a, b, c);
}
struct HullFaces { const float4* verts; const uint32_t* ranges; const uint32_t* tris; const uint32_t* triRanges; };
// One collider in its entity's local frame; `s0..s2` = the collider_desc shape words.
__device__ inline bool rayVsCollider(uint32_t type, float4 s0, float4 s1, float4 s2, const HullFaces& hf, V3 o, V3 d, float& t) {
    switch (type) {
        case T_SPHERE: return raySphere(o, d, xyz(s0), s0.w, t);
        case T_CAPSULE: {
            V3 pa = xyz(s0), pb(s0.w, s1.x, s1.y); float r = s1.z;
            t = FLT_MAX;
            float tt; bool result = false;
            if (rayCylinder(o, d, pa, pb, r, tt)) { t = tt; result = true; }
            if (raySphere(o, d, pa, r, tt)) { t = fminr(t, tt); result = true; }
            if (raySphere(o, d, pb, r, tt)) { t = fminr(t, tt); result = true; }
            return result;
        }
        case T_CYLINDER: return rayCylinder(o, d, xyz(s0), V3(s0.w, s1.x, s1.y), s1.z, t);
        case T_AABB: return rayAABB(o, d, xyz(s0), V3(s0.w, s1.x, s1.y), t);
        case T_OBB: {
            Q4 inv = conj(Q4(s0.x, s0.y, s0.z, s0.w)); V3 c(s1.x, s1.y, s1.z), r(s1.w, s2.x, s2.y);
            return rayAABB(rotate(inv, o - c), rotate(inv, d), V3() - r, V3() + r, t);
        }
        default: {
            Q4 inv = conj(Q4(s0.x, s0.y, s0.z, s0.w)); V3 pos(s1.x, s1.y, s1.z);
            const uint32_t geom = __float_as_uint(s1.w);
            V3 lo = rotate(inv, o - pos), ld = rotate(inv, d);
            const uint32_t v0 = hf.ranges[2 * geom], f0 = hf.triRanges[2 * geom], nf = hf.triRanges[2 * geom + 1];
            float minT = FLT_MAX; bool result = false;
            for (uint32_t f = 0; f < nf; ++f) {
                float tt;
                V3 a = xyz(hf.verts[v0 + hf.tris[3 * (f0 + f)]]), b = xyz(hf.verts[v0 + hf.tris[3 * (f0 + f) + 1]]), c = xyz(hf.verts[v0 + hf.tris[3 * (f0 + f) + 2]]);
                if (rayTriangle(lo, ld, a, b, c, tt) && tt < minT) { minT = tt; result = true; }
            }
            t = minT;
            return result;
        }
    }
}
// One workgroup per ray over all colliders (world index order = the reference's view order; the first collider with the
// smallest t wins: strict `<`).  Output per ray: body (or ~0u) and (force, torque) for k_add_forces.
__global__ __launch_bounds__(256) void k_ray_interactions(uint32_t nc, const float* __restrict__ rays /* origin3, direction3, strength, - */, const uint32_t* __restrict__ ranges,
                                                          const uint32_t* __restrict__ cTypeBody, const uint32_t* __restrict__ cEntity, const float4* __restrict__ cShape,
                                                          const float4* __restrict__ bPos, const float4* __restrict__ bRot, const float4* __restrict__ bCogInvMass,
                                                          HullFaces hf, uint32_t* __restrict__ outBody, float* __restrict__ outFT) {
    __shared__ unsigned long long best[256];
    const uint32_t r = blockIdx.x;
    const V3 ro(rays[8 * r], rays[8 * r + 1], rays[8 * r + 2]), rd(rays[8 * r + 3], rays[8 * r + 4], rays[8 * r + 5]);
    const float strength = rays[8 * r + 6];
    const uint32_t lo = ranges[2 * r], hi = ranges[2 * r + 1];
    unsigned long long mine = ~0ull;   // (t bits << 32 | collider): t >= 0, so the bit pattern orders like the value
    for (uint32_t k = threadIdx.x; k < nc; k += blockDim.x) {
        const uint32_t body = cTypeBody[2 * k + 1], ent = cEntity[k];
        if (body == kNoBody || ent < lo || ent >= hi) continue;
        const Q4 inv = conj(toQ(bRot[body]));
        float t;
        if (rayVsCollider(cTypeBody[2 * k], cShape[3 * k], cShape[3 * k + 1], cShape[3 * k + 2], hf, rotate(inv, ro - xyz(bPos[body])), rotate(inv, rd), t) && t < FLT_MAX) {
            unsigned long long key = ((unsigned long long)__float_as_uint(t + 0.f) << 32) | k;   // -0 -> +0
            if (key < mine) mine = key;
        }
    }
    best[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) { if (threadIdx.x < s && best[threadIdx.x + s] < best[threadIdx.x]) best[threadIdx.x] = best[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x != 0) return;
    if (best[0] == ~0ull) { outBody[r] = 0xFFFFFFFFu; return; }
    const uint32_t k = (uint32_t)best[0], body = cTypeBody[2 * k + 1];
    const float t = __uint_as_float((uint32_t)(best[0] >> 32));
    const Q4 rot = toQ(bRot[body]); const V3 pos = xyz(bPos[body]);
    const V3 lo_ = rotate(conj(rot), ro - pos), ld = rotate(conj(rot), rd);
    const V3 globalHit = rotate(rot, lo_ + t * ld) + pos;
    const V3 cog = pos + rotate(rot, xyz(bCogInvMass[body]));
    const V3 force = rd * strength, torque = cross(globalHit - cog, force);
    outBody[r] = body;
    outFT[6 * r] = force.x; outFT[6 * r + 1] = force.y; outFT[6 * r + 2] = force.z; outFT[6 * r + 3] = torque.x; outFT[6 * r + 4] = torque.y; outFT[6 * r + 5] = torque.z;
}

// ---- EPA (collision_epa.h:96-168, collision_epa.cpp)
constexpr int kEpaPts = 24, kEpaTris = 288, kEpaEdges = 288, kEpaBorder = 32;
struct EpaTri { uint16_t a, b, c, eA, eB, eC; V3 n; float dist; };
struct EpaEdge { uint16_t a, b, tA, tB; };
struct EpaState {
    SupPt pts[kEpaPts];
    EpaTri tris[kEpaTris];
    EpaEdge edges[kEpaEdges];
    uint8_t active[kEpaTris];
    uint8_t refs[kEpaEdges];
    uint16_t nTris, nPts, nEdges;
};

__device__ __forceinline__ void triInfo(const SupPt& a, const SupPt& b, const SupPt& c, V3& n, float& dist) {
    n = normalize(cross(b.m - a.m, c.m - a.m));
    dist = dot(n, a.m);
}
__device__ __forceinline__ uint16_t epaPushPt(EpaState& s, const SupPt& p) {
    if (s.nPts >= kEpaPts) return 0xFFFF;
    s.pts[s.nPts] = p; return s.nPts++;
}
__device__ __forceinline__ uint16_t epaPushTri(EpaState& s, uint16_t a, uint16_t b, uint16_t c, uint16_t eA, uint16_t eB, uint16_t eC, V3 n, float dist) {
    if (s.nTris >= kEpaTris) return 0xFFFF;
    uint16_t i = s.nTris++;
    s.active[i] = 1;
    EpaTri& t = s.tris[i];
    t.a = a; t.b = b; t.c = c; t.eA = eA; t.eB = eB; t.eC = eC; t.n = n; t.dist = dist;
    return i;
}
__device__ __forceinline__ uint16_t epaPushEdge(EpaState& s, uint16_t a, uint16_t b, uint16_t tA, uint16_t tB) {
    if (s.nEdges >= kEpaEdges) return 0xFFFF;
    uint16_t i = s.nEdges++;
    EpaEdge e; e.a = a; e.b = b; e.tA = tA; e.tB = tB;
    s.edges[i] = e;
    return i;
}
// Where the reference's code reads what it never wrote — on a degenerate polytope (zero-area faces have NaN normals and are never "seen", so the faces a new point sees need
// not form one loop) a new face's third edge is looked up in newEdgePerPoint[] for a point no horizon edge started from, and an edge's first face may stay unset — the
// reference's behaviour is undefined (it indexes its 1024-entry arrays with whatever the stack held).  Product and oracle (oracle/ora_gjk.cpp) define it the same way: such
// an index is 0xFFFF, "no edge" / "no face": it is never dereferenced, counts no reference, and a missing face counts as inactive.  (No index leaves the polytope's arrays.)
__device__ __forceinline__ bool epaEdgeIdx(uint16_t e) { return e < (uint16_t)kEpaEdges; }
__device__ __forceinline__ bool epaTriIdx(uint16_t t) { return t < (uint16_t)kEpaTris; }
__device__ inline bool epaAddPoint(EpaState& s, const SupPt& np) {  // collision_epa.cpp:117-240
    for (uint32_t i = 0; i < s.nEdges; ++i) s.refs[i] = 0;
    for (uint32_t i = 0; i < s.nTris; ++i) {
        if (!s.active[i]) continue;
        EpaTri& t = s.tris[i];
        float d = dot(t.n, np.m - s.pts[t.a].m);
        if (d > 0.f) { if (epaEdgeIdx(t.eA)) ++s.refs[t.eA]; if (epaEdgeIdx(t.eB)) ++s.refs[t.eB]; if (epaEdgeIdx(t.eC)) ++s.refs[t.eC]; s.active[i] = 0; }
    }
    uint16_t border[kEpaBorder]; uint32_t nb = 0;
    for (uint32_t i = 0; i < s.nEdges; ++i)
        if (s.refs[i] == 1) { if (nb >= (uint32_t)kEpaBorder) return false; border[nb++] = (uint16_t)i; }
    uint16_t newEdgePerPoint[kEpaPts];
    for (int i = 0; i < kEpaPts; ++i) newEdgePerPoint[i] = 0xFFFF;
    uint16_t npi = epaPushPt(s, np);
    if (npi == 0xFFFF) return false;
    uint16_t triOffset = s.nTris;
    for (uint32_t i = 0; i < nb; ++i) {
        uint16_t ei = border[i];
        EpaEdge e = s.edges[ei];
        bool aAct = epaTriIdx(e.tA) && s.active[e.tA] != 0, bAct = epaTriIdx(e.tB) && s.active[e.tB] != 0;
        uint16_t connect = bAct ? e.a : e.b;
        uint16_t triIndex = s.nTris;
        uint16_t ne = epaPushEdge(s, connect, npi, 0xFFFF, s.nTris);
        if (ne == 0xFFFF) return false;
        newEdgePerPoint[connect] = ne;
        uint16_t bI = connect, cI = bAct ? e.b : e.a;
        V3 n; float dist;
        triInfo(np, s.pts[bI], s.pts[cI], n, dist);
        uint16_t test = epaPushTri(s, npi, bI, cI, ei, 0xFFFF, ne, n, dist);
        if (test == 0xFFFF) return false;
        if (aAct) s.edges[ei].tB = triIndex; else s.edges[ei].tA = triIndex;
    }
    for (uint32_t i = 0; i < nb; ++i) {
        EpaEdge e = s.edges[border[i]];
        bool bNew = e.tB >= triOffset;
        uint16_t connect = bNew ? e.a : e.b;
        uint16_t other = newEdgePerPoint[connect];
        uint16_t triIndex = (uint16_t)(i + triOffset);
        s.tris[triIndex].eB = other;
        if (epaEdgeIdx(other)) s.edges[other].tA = triIndex;
    }
    return true;
}
__device__ __forceinline__ V3 barycentric(V3 a, V3 b, V3 c, V3 p) {  // src/core/math.cpp:1391-1407
    V3 v0 = b - a, v1 = c - a, v2 = p - a;
    float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    denom = (fabsf(denom) < kEps) ? 1.f : denom;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    return V3(u, v, w);
}
struct EpaOut { V3 point, normal; float depth; };

__device__ inline void epaRun(const Simplex& g, const Shape& A, const Shape& B, const HullSet& hs, EpaState& s, EpaOut& out) {
    s.nTris = 0; s.nPts = 0; s.nEdges = 0;
    epaPushPt(s, g.a); epaPushPt(s, g.b); epaPushPt(s, g.c); epaPushPt(s, g.d);
    V3 n; float dist;
    triInfo(g.a, g.b, g.d, n, dist); epaPushTri(s, 0, 1, 3, 4, 3, 0, n, dist);
    triInfo(g.b, g.c, g.d, n, dist); epaPushTri(s, 1, 2, 3, 5, 4, 1, n, dist);
    triInfo(g.c, g.a, g.d, n, dist); epaPushTri(s, 2, 0, 3, 3, 5, 2, n, dist);
    triInfo(g.a, g.c, g.b, n, dist); epaPushTri(s, 0, 2, 1, 1, 0, 2, n, dist);
    epaPushEdge(s, 0, 1, 0, 3); epaPushEdge(s, 1, 2, 1, 3); epaPushEdge(s, 2, 0, 2, 3);
    epaPushEdge(s, 0, 3, 2, 0); epaPushEdge(s, 1, 3, 0, 1); epaPushEdge(s, 2, 3, 1, 2);
    uint32_t closest = 0;
    for (uint32_t it = 0; it < 20; ++it) {
        uint32_t prev = closest;
        closest = 0xFFFFFFFFu; float minD = FLT_MAX;
        for (uint32_t i = 0; i < s.nTris; ++i)
            if (s.active[i] && s.tris[i].dist < minD) { minD = s.tris[i].dist; closest = i; }
        if (closest == 0xFFFFFFFFu) { closest = prev; break; }   // degenerate polytope: keep the last face (the reference asserts here)
        V3 tn = s.tris[closest].n; float td = s.tris[closest].dist;
        SupPt a = supportPair(A, B, hs, tn);
        float d = dot(a.m, tn);
        if (d - td < 0.01f) break;
        if (!epaAddPoint(s, a)) break;
    }
    const EpaTri& t = s.tris[closest];
    const SupPt& a = s.pts[t.a]; const SupPt& b = s.pts[t.b]; const SupPt& c = s.pts[t.c];
    V3 bc = barycentric(a.m, b.m, c.m, t.n * t.dist);
    V3 pA = bc.x * a.a + bc.y * b.a + bc.z * c.a;
    V3 pB = bc.x * a.b + bc.y * b.b + bc.z * c.b;
    out.point = 0.5f * (pA + pB);
    out.normal = t.n;
    out.depth = t.dist;
}

// ---- EPA by ONE WAVE per pair (k_narrow_epa).  The polytope lives in LDS; the lanes share the three scans that make a per-lane EPA
// slow — closest face, faces the new point can see, hull vertices of a support query — and the arithmetic of every face / vertex is
// the very same sequence of operations, so is what they pick: minima and maxima keep the FIRST best element of the sequential scan
// (ties go to the lower index), horizon edges are taken in ascending edge order.  The few order-sensitive link updates stay with lane 0.
struct EpaLds {
    SupPt pts[kEpaPts];
    EpaTri tris[kEpaTris];
    EpaEdge edges[kEpaEdges];
    uint32_t refs[kEpaEdges];
    uint8_t active[kEpaTris];
    uint16_t border[kEpaBorder];
    uint16_t newEdgePerPoint[kEpaPts];
    float bestD[2]; uint32_t bestI[2];
};
__device__ __forceinline__ void waveSync() { __syncthreads(); }   // one wave per workgroup: orders its LDS traffic
__device__ inline V3 supportOfWave(const Shape& s, const HullSet& hs, V3 dir, uint32_t lane) {
    if (s.type != T_HULL) return supportOf(s, hs, dir);
    dir = rotate(conj(s.rot), dir);
    const uint32_t first = hs.ranges[2 * s.hull], count = hs.ranges[2 * s.hull + 1];
    float maxD = -FLT_MAX; uint32_t bestI = 0xFFFFFFFFu;
    for (uint32_t i = lane; i < count; i += 64u) {
        const float d = dot(dir, xyz(hs.verts[first + i]));
        if (d > maxD) { maxD = d; bestI = i; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {   // first maximum of the sequential scan = largest value, lowest index among equals
        const float od = __shfl_xor(maxD, off, 64); const uint32_t oi = (uint32_t)__shfl_xor((int)bestI, off, 64);
        if (oi != 0xFFFFFFFFu && (bestI == 0xFFFFFFFFu || od > maxD || (od == maxD && oi < bestI))) { maxD = od; bestI = oi; }
    }
    V3 best;   // (no vertex beat -FLT_MAX: the sequential code leaves `best` unset; hull vertices are finite, this does not happen)
    if (bestI != 0xFFFFFFFFu) best = xyz(hs.verts[first + bestI]);
    return s.a + rotate(s.rot, best);
}
__device__ inline SupPt supportPairWave(const Shape& A, const Shape& B, const HullSet& hs, V3 dir, uint32_t lane) {
    SupPt p;
    p.a = supportOfWave(A, hs, dir, lane);
    p.b = supportOfWave(B, hs, -dir, lane);
    p.m = p.a - p.b;
    return p;
}
__device__ inline bool epaAddPointWave(EpaLds& s, uint32_t& nTris, uint32_t& nPts, uint32_t& nEdges, const SupPt& np, uint32_t lane) {  // = epaAddPoint
    for (uint32_t i = lane; i < nEdges; i += 64u) s.refs[i] = 0u;
    if (lane < (uint32_t)kEpaPts) s.newEdgePerPoint[lane] = 0xFFFF;
    waveSync();
    for (uint32_t i = lane; i < nTris; i += 64u) {
        if (!s.active[i]) continue;
        const EpaTri t = s.tris[i];
        const float d = dot(t.n, np.m - s.pts[t.a].m);
        if (d > 0.f) { if (epaEdgeIdx(t.eA)) atomicAdd(&s.refs[t.eA], 1u); if (epaEdgeIdx(t.eB)) atomicAdd(&s.refs[t.eB], 1u); if (epaEdgeIdx(t.eC)) atomicAdd(&s.refs[t.eC], 1u); s.active[i] = 0; }
    }
    waveSync();
    uint32_t nb = 0;   // horizon = edges referenced once, in ascending edge order
    for (uint32_t base = 0; base < nEdges; base += 64u) {
        const uint32_t i = base + lane;
        const bool flag = i < nEdges && s.refs[i] == 1u;
        const unsigned long long mask = __ballot(flag);
        if (flag) { const uint32_t pos = nb + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)); if (pos < (uint32_t)kEpaBorder) s.border[pos] = (uint16_t)i; }
        nb += (uint32_t)__popcll(mask);
    }
    if (nb > (uint32_t)kEpaBorder) return false;
    if (nPts >= (uint32_t)kEpaPts) return false;
    const uint16_t npi = (uint16_t)nPts;
    if (lane == 0) s.pts[nPts] = np;
    ++nPts;
    const uint32_t triOffset = nTris, edgeOffset = nEdges;
    // (the sequential code stops at the first push that does not fit and leaves the polytope half-extended; the caller then ends the
    // iteration and only reads the face it had chosen before, which nothing here touches — so the check can come first)
    if (edgeOffset + nb > (uint32_t)kEpaEdges || triOffset + nb > (uint32_t)kEpaTris) return false;
    waveSync();
    if (lane < nb) {
        const uint16_t ei = s.border[lane];
        const EpaEdge e = s.edges[ei];
        const bool bAct = epaTriIdx(e.tB) && s.active[e.tB] != 0;
        const uint16_t connect = bAct ? e.a : e.b;
        const uint16_t triIndex = (uint16_t)(triOffset + lane), ne = (uint16_t)(edgeOffset + lane);
        EpaEdge nw; nw.a = connect; nw.b = npi; nw.tA = 0xFFFF; nw.tB = triIndex;
        s.edges[ne] = nw;
        const uint16_t bI = connect, cI = bAct ? e.b : e.a;
        V3 n; float dist;
        triInfo(np, s.pts[bI], s.pts[cI], n, dist);
        EpaTri t; t.a = npi; t.b = bI; t.c = cI; t.eA = ei; t.eB = 0xFFFF; t.eC = ne; t.n = n; t.dist = dist;
        s.tris[triIndex] = t;
    }
    waveSync();
    // The link updates of both sequential loops are order-sensitive only if a point is the `connect` of two horizon edges (a degenerate
    // polytope): with distinct points — a proper horizon loop — every lane links its own edge, else lane 0 replays the loops in order.
    uint32_t myConnect = 0xFFFFFFFFu, myOther = 0xFFFFFFFFu;   // the edge's end the new edge starts from (first loop) / the end whose new edge closes the face (second loop)
    if (lane < nb) { const EpaEdge e = s.edges[s.border[lane]]; const bool aAct = epaTriIdx(e.tA) && s.active[e.tA] != 0, bAct = epaTriIdx(e.tB) && s.active[e.tB] != 0; myConnect = bAct ? e.a : e.b; myOther = aAct ? e.a : e.b; }
    unsigned int pointBits = lane < nb ? (1u << myConnect) : 0u, otherBits = lane < nb ? (1u << myOther) : 0u;   // kEpaPts = 24 points
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { pointBits |= (unsigned int)__shfl_xor((int)pointBits, off, 64); otherBits |= (unsigned int)__shfl_xor((int)otherBits, off, 64); }
    const bool distinct = (uint32_t)__popc(pointBits) == nb && (uint32_t)__popc(otherBits) == nb;
    if (distinct) {
        if (lane < nb) {
            const uint16_t ei = s.border[lane];
            const EpaEdge e = s.edges[ei];
            const bool aAct = epaTriIdx(e.tA) && s.active[e.tA] != 0;
            const uint16_t triIndex = (uint16_t)(triOffset + lane);
            s.newEdgePerPoint[myConnect] = (uint16_t)(edgeOffset + lane);
            if (aAct) s.edges[ei].tB = triIndex; else s.edges[ei].tA = triIndex;
        }
        waveSync();
        if (lane < nb) {
            s.active[triOffset + lane] = 1;
            const EpaEdge e = s.edges[s.border[lane]];
            const bool bNew = e.tB >= triOffset;
            const uint16_t connect = bNew ? e.a : e.b;
            const uint16_t other = s.newEdgePerPoint[connect];
            const uint16_t triIndex = (uint16_t)(lane + triOffset);
            s.tris[triIndex].eB = other;
            if (epaEdgeIdx(other)) s.edges[other].tA = triIndex;
        }
    } else if (lane == 0) {
        for (uint32_t i = 0; i < nb; ++i) {
            const uint16_t ei = s.border[i];
            const EpaEdge e = s.edges[ei];
            const bool aAct = epaTriIdx(e.tA) && s.active[e.tA] != 0, bAct = epaTriIdx(e.tB) && s.active[e.tB] != 0;
            const uint16_t connect = bAct ? e.a : e.b;
            const uint16_t triIndex = (uint16_t)(triOffset + i);
            s.newEdgePerPoint[connect] = (uint16_t)(edgeOffset + i);
            s.active[triIndex] = 1;
            if (aAct) s.edges[ei].tB = triIndex; else s.edges[ei].tA = triIndex;
        }
        for (uint32_t i = 0; i < nb; ++i) {
            const EpaEdge e = s.edges[s.border[i]];
            const bool bNew = e.tB >= triOffset;
            const uint16_t connect = bNew ? e.a : e.b;
            const uint16_t other = s.newEdgePerPoint[connect];
            const uint16_t triIndex = (uint16_t)(i + triOffset);
            s.tris[triIndex].eB = other;
            if (epaEdgeIdx(other)) s.edges[other].tA = triIndex;
        }
    }
    nTris += nb; nEdges += nb;
    waveSync();
    return true;
}
__device__ inline void epaRunWave(const Simplex& g, const Shape& A, const Shape& B, const HullSet& hs, EpaLds& s, EpaOut& out, uint32_t lane) {  // = epaRun
    uint32_t nTris = 4, nPts = 4, nEdges = 6;
    if (lane == 0) {
        s.pts[0] = g.a; s.pts[1] = g.b; s.pts[2] = g.c; s.pts[3] = g.d;
        auto tri = [&](uint32_t i, const SupPt& a, const SupPt& b, const SupPt& c, uint16_t ia, uint16_t ib, uint16_t ic, uint16_t eA, uint16_t eB, uint16_t eC) {
            V3 n; float dist; triInfo(a, b, c, n, dist);
            EpaTri t; t.a = ia; t.b = ib; t.c = ic; t.eA = eA; t.eB = eB; t.eC = eC; t.n = n; t.dist = dist;
            s.tris[i] = t; s.active[i] = 1;
        };
        tri(0, g.a, g.b, g.d, 0, 1, 3, 4, 3, 0);
        tri(1, g.b, g.c, g.d, 1, 2, 3, 5, 4, 1);
        tri(2, g.c, g.a, g.d, 2, 0, 3, 3, 5, 2);
        tri(3, g.a, g.c, g.b, 0, 2, 1, 1, 0, 2);
        auto edge = [&](uint32_t i, uint16_t a, uint16_t b, uint16_t tA, uint16_t tB) { EpaEdge e; e.a = a; e.b = b; e.tA = tA; e.tB = tB; s.edges[i] = e; };
        edge(0, 0, 1, 0, 3); edge(1, 1, 2, 1, 3); edge(2, 2, 0, 2, 3);
        edge(3, 0, 3, 2, 0); edge(4, 1, 3, 0, 1); edge(5, 2, 3, 1, 2);
    }
    waveSync();
    uint32_t closest = 0;
    for (uint32_t it = 0; it < 20; ++it) {
        const uint32_t prev = closest;
        float minD = FLT_MAX; uint32_t best = 0xFFFFFFFFu;   // first minimum of the sequential scan
        for (uint32_t i = lane; i < nTris; i += 64u)
            if (s.active[i]) { const float d = s.tris[i].dist; if (d < minD) { minD = d; best = i; } }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(minD, off, 64); const uint32_t oi = (uint32_t)__shfl_xor((int)best, off, 64);
            if (oi != 0xFFFFFFFFu && (best == 0xFFFFFFFFu || od < minD || (od == minD && oi < best))) { minD = od; best = oi; }
        }
        closest = best;
        if (closest == 0xFFFFFFFFu) { closest = prev; break; }
        const V3 tn = s.tris[closest].n; const float td = s.tris[closest].dist;
        const SupPt a = supportPairWave(A, B, hs, tn, lane);
        const float d = dot(a.m, tn);
        if (d - td < 0.01f) break;
        if (!epaAddPointWave(s, nTris, nPts, nEdges, a, lane)) break;
    }
    const EpaTri t = s.tris[closest];
    const SupPt a = s.pts[t.a], b = s.pts[t.b], c = s.pts[t.c];
    V3 bc = barycentric(a.m, b.m, c.m, t.n * t.dist);
    V3 pA = bc.x * a.a + bc.y * b.a + bc.z * c.a;
    V3 pB = bc.x * a.b + bc.y * b.b + bc.z * c.b;
    out.point = 0.5f * (pA + pB);
    out.normal = t.n;
    out.depth = t.dist;
}

// The second half of every GJK+EPA test, from the simplex GJK ended with: penetration by EPA -> one contact.
__device__ inline void epaSingle(const Simplex& sx, const Shape& a, const Shape& b, const HullSet& hs, EpaState& st, Manifold& out, EpaOut& epa) {
    epaRun(sx, a, b, hs, st, epa);   // EPA status is ignored by every caller (collision_narrow.cpp:509-512)
    out.n = epa.normal;
    out.count = 1;
    out.d[0] = epa.depth;
    out.p[0] = epa.point;
}
__device__ inline bool gjkEpaSingle(const Shape& a, const Shape& b, const HullSet& hs, EpaState& st, Manifold& out, EpaOut& epa) {
    Simplex sx;
    if (!gjkTest(a, b, hs, sx)) return false;
    epaSingle(sx, a, b, hs, st, out, epa);
    return true;
}

// capsule / cylinder vs AABB (collision_narrow.cpp:705-768, 953-1020): if EPA found a box-face normal and the
// segment is parallel to that face, clip the segment against the face -> up to 2 contacts.
__device__ inline void segmentShapeVsAABBAfterEpa(const Shape& c, const Shape& box, const EpaOut& epa, Manifold& out);
__device__ inline bool segmentShapeVsAABB(const Shape& c, const Shape& box, const HullSet& hs, EpaState& st, Manifold& out) {
    EpaOut epa;
    if (!gjkEpaSingle(c, box, hs, st, out, epa)) return false;
    segmentShapeVsAABBAfterEpa(c, box, epa, out);
    return true;
}
__device__ inline void segmentShapeVsAABBAfterEpa(const Shape& c, const Shape& box, const EpaOut& epa, Manifold& out) {
    V3 normal = epa.normal;
    if (fabsf(normal.x) > 0.99f || fabsf(normal.y) > 0.99f || fabsf(normal.z) > 0.99f) {
        V3 axis = normalize(c.b - c.a);
        if (fabsf(dot(normal, axis)) < 0.01f) {
            V3 cp[4], cn[4]; P4 planes[4];
            V3 boxNormal = -normal;
            P4 ref = boxReferencePlane(box.a, box.b, boxNormal);
            ClipPoly poly; poly.n = 2;
            V3 pa = c.a + normal * c.radius, pb = c.b + normal * c.radius;
            poly.pt[0].v = pa; poly.pt[0].depth = -planeDist(pa, ref);
            poly.pt[1].v = pb; poly.pt[1].depth = -planeDist(pb, ref);
            V3 center = (box.a + box.b) * 0.5f;
            boxClipPlanes((box.b - box.a) * 0.5f, boxNormal, cp, cn);
            for (int i = 0; i < 4; ++i) { cp[i] = cp[i] + center; planes[i] = makePlane(cp[i], cn[i]); }
            ClipPoly clipped;
            clipAndBuild(poly, clipped, planes, 4, ref, out);
        }
    }
}

// cylinder vs cylinder, the closed-form case of (nearly) parallel axes (collision_narrow.cpp:821-925).  Returns false when the
// axes are not parallel (GJK + EPA decide then); otherwise `hit` / `out` are final.
__device__ inline bool cylinderCylinderParallel(const Shape& a, const Shape& b, bool& hit, Manifold& out) {
    V3 aDir = a.b - a.a;
    V3 bDir = normalize(b.b - b.a);
    float aLen = len(aDir);
    aDir = aDir * (1.f / aLen);
    float parallel = dot(aDir, bDir);
    if (fabsf(parallel) > 0.99f) {
        V3 pBa = b.a, pBb = b.b;
        if (parallel < 0.f) { V3 t = pBa; pBa = pBb; pBb = t; }
        V3 ref = a.a;
        float a0 = 0.f, a1 = aLen;
        float b0 = dot(aDir, pBa - ref), b1 = dot(aDir, pBb - ref);
        float left = fmaxr(a0, b0), right = fminr(a1, b1);
        hit = false;
        if (right < left) return true;
        V3 cA0 = ref + left * aDir, cA1 = ref + right * aDir;
        V3 cB0 = closestOnSegment(cA0, pBa, pBb);
        V3 cB1 = cB0 + (right - left) * aDir;
        V3 normal = cB0 - cA0;
        float d = len(normal);
        float pen = (a.radius + b.radius) - d;
        if (pen < 0.f) return true;
        hit = true;
        float capPen = right - left;
        if (capPen < pen) {
            out.count = 1;
            out.d[0] = capPen;
            // the reference applies the scalar to every component here (vec3 -/+ float): kept as written
            if (b0 > a0) { out.n = aDir; out.p[0] = a.b - V3(capPen * 0.5f); }
            else { out.n = -aDir; out.p[0] = a.a + V3(capPen * 0.5f); }
        } else {
            if (d < kEps) { d = 0.f; normal = V3(0.f, 1.f, 0.f); }
            else normal = normal / d;
            out.n = normal;
            out.count = 2;
            out.d[0] = pen; out.p[0] = (cA0 + cB0) * 0.5f;
            out.d[1] = pen; out.p[1] = (cA1 + cB1) * 0.5f;
        }
        return true;
    }
    return false;
}
__device__ inline bool cylinderCylinder(const Shape& a, const Shape& b, const HullSet& hs, EpaState& st, Manifold& out) {  // 821-951
    bool hit;
    if (cylinderCylinderParallel(a, b, hit, out)) return hit;
    EpaOut epa;
    return gjkEpaSingle(a, b, hs, st, out, epa);
}

// mode: 0 plain GJK+EPA single contact, 1 segment shape vs AABB, 2 segment shape vs OBB, 3 cylinder vs cylinder
__device__ inline bool intersectGjkImpl(const Shape& a, const Shape& b, const HullSet& hs, EpaState& st, Manifold& out, int mode) {
    EpaOut epa;
    switch (mode) {
        case 1: return segmentShapeVsAABB(a, b, hs, st, out);
        case 2: {  // into the box frame and back (770-790, 1022-1043)
            Shape c = a;
            c.a = rotate(conj(b.rot), a.a - b.a) + b.a;
            c.b = rotate(conj(b.rot), a.b - b.a) + b.a;
            Shape box; box.type = T_AABB; box.a = b.a - b.b; box.b = b.a + b.b; box.radius = 0.f; box.hull = 0;
            if (!segmentShapeVsAABB(c, box, hs, st, out)) return false;
            out.n = rotate(b.rot, out.n);
            for (uint32_t i = 0; i < out.count; ++i) out.p[i] = rotate(b.rot, out.p[i] - b.a) + b.a;
            return true;
        }
        case 3: return cylinderCylinder(a, b, hs, st, out);
        default: return gjkEpaSingle(a, b, hs, st, out, epa);
    }
}

// GJK / EPA in two kernels.  k_narrow_gjk: one LANE per pair of the GJK buckets — the shape preparation, the closed-form parallel-cylinder
// case and GJK itself (a handful of iterations); a pair whose shapes intersect goes, with its final simplex, into the EPA queue.
// k_narrow_epa: one WAVE per queued pair — EPA with the polytope in LDS and its scans shared by the lanes (epaRunWave), then the
// shape-specific contact construction.  (One lane per pair for EPA as well made the whole kernel as slow as its slowest lane: ~100 us
// for a single pair that runs all 20 iterations with ~12 KB of per-lane scratch, on a chip the few thousand pairs leave idle.)
// gjkPhase: 0 = no contact, 1 = `out` is final, 2 = intersecting, EPA to follow from `sx`
__device__ inline int gjkPhase(const Shape& a, const Shape& b, const HullSet& hs, int mode, Simplex& sx, Manifold& out) {
    switch (mode) {
        case 2: {  // into the box frame (770-790, 1022-1043)
            Shape c = a;
            c.a = rotate(conj(b.rot), a.a - b.a) + b.a;
            c.b = rotate(conj(b.rot), a.b - b.a) + b.a;
            Shape box; box.type = T_AABB; box.a = b.a - b.b; box.b = b.a + b.b; box.radius = 0.f; box.hull = 0;
            return gjkTest(c, box, hs, sx) ? 2 : 0;
        }
        case 3: {
            bool hit;
            if (cylinderCylinderParallel(a, b, hit, out)) return hit ? 1 : 0;
            return gjkTest(a, b, hs, sx) ? 2 : 0;
        }
        default: return gjkTest(a, b, hs, sx) ? 2 : 0;
    }
}
// = the EPA half of intersectGjkImpl, by one wave (every lane ends with the same `out`)
__device__ inline void epaPhaseWave(const Shape& a, const Shape& b, const HullSet& hs, int mode, const Simplex& sx, EpaLds& lds, Manifold& out, uint32_t lane) {
    EpaOut epa;
    auto single = [&]() { out.n = epa.normal; out.count = 1; out.d[0] = epa.depth; out.p[0] = epa.point; };
    switch (mode) {
        case 1: epaRunWave(sx, a, b, hs, lds, epa, lane); single(); segmentShapeVsAABBAfterEpa(a, b, epa, out); break;
        case 2: {
            Shape c = a;
            c.a = rotate(conj(b.rot), a.a - b.a) + b.a;
            c.b = rotate(conj(b.rot), a.b - b.a) + b.a;
            Shape box; box.type = T_AABB; box.a = b.a - b.b; box.b = b.a + b.b; box.radius = 0.f; box.hull = 0;
            epaRunWave(sx, c, box, hs, lds, epa, lane); single(); segmentShapeVsAABBAfterEpa(c, box, epa, out);
            out.n = rotate(b.rot, out.n);
            for (uint32_t i = 0; i < out.count; ++i) out.p[i] = rotate(b.rot, out.p[i] - b.a) + b.a;
        } break;
        default: epaRunWave(sx, a, b, hs, lds, epa, lane); single(); break;
    }
}
__device__ __forceinline__ int gjkPairShapes(uint64_t key, const float4* __restrict__ wShape, Shape& sa, Shape& sb) {
    uint32_t bucket = (uint32_t)(key >> 58), a = (uint32_t)((key >> 29) & 0x1FFFFFFFu), b = (uint32_t)(key & 0x1FFFFFFFu);
    uint32_t ta = 0, rem = bucket;
    while (rem >= 6u - ta) { rem -= 6u - ta; ++ta; }
    uint32_t tb = ta + rem;
    int mode = gjkMode(ta, tb);
    if (mode < 0) return mode;
    sa = loadShape(wShape, a, ta); sb = loadShape(wShape, b, tb);
    return mode;
}
// gjkTest / gjkPhase with the support queries shared by the lanes of a wave (hull vertices), everything else computed by every lane alike
__device__ inline bool gjkTestWave(const Shape& A, const Shape& B, const HullSet& hs, Simplex& sx, uint32_t lane) {
    V3 dir(1.f, 0.1f, -0.2f);
    sx.n = 0;
    sx.c = supportPairWave(A, B, hs, dir, lane);
    if (dot(sx.c.m, dir) < 0.f) return false;
    dir = -sx.c.m;
    sx.b = supportPairWave(A, B, hs, dir, lane);
    if (dot(sx.b.m, dir) < 0.f) return false;
    dir = crossABA(sx.c.m - sx.b.m, -sx.b.m);
    sx.n = 2;
    for (int guard = 0; guard < 64; ++guard) {
        if (sqlen(dir) < 0.0001f) return false;
        SupPt a = supportPairWave(A, B, hs, dir, lane);
        if (dot(a.m, dir) < 0.f) return false;
        int r = updateSimplex(sx, a, dir);
        if (r == GJK_STOP) { sx.a = a; sx.n = 4; return true; }
        if (r == GJK_ERR) return false;
    }
    return false;
}
__device__ inline int gjkPhaseWave(const Shape& a, const Shape& b, const HullSet& hs, int mode, Simplex& sx, Manifold& out, uint32_t lane) {
    switch (mode) {
        case 2: {
            Shape c = a;
            c.a = rotate(conj(b.rot), a.a - b.a) + b.a;
            c.b = rotate(conj(b.rot), a.b - b.a) + b.a;
            Shape box; box.type = T_AABB; box.a = b.a - b.b; box.b = b.a + b.b; box.radius = 0.f; box.hull = 0;
            return gjkTestWave(c, box, hs, sx, lane) ? 2 : 0;
        }
        case 3: {
            bool hit;
            if (cylinderCylinderParallel(a, b, hit, out)) return hit ? 1 : 0;
            return gjkTestWave(a, b, hs, sx, lane) ? 2 : 0;
        }
        default: return gjkTestWave(a, b, hs, sx, lane) ? 2 : 0;
    }
}
// GJK and EPA of a pair by one wave, no queue: the variant for FEW pairs (a lane-per-pair GJK lasts as long as its slowest lane)
__global__ __launch_bounds__(64) void k_narrow_gjk_wave(const StepScalars* __restrict__ sc, const uint64_t* __restrict__ pairsA, const uint64_t* __restrict__ pairsB,
                                                        const float4* __restrict__ wShape, HullSet hs, uint64_t* __restrict__ npPacked, float4* __restrict__ npNormal, float4* __restrict__ npPoints) {
    __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[sizeof(EpaLds)];
    EpaLds& lds = *reinterpret_cast<EpaLds*>(ldsRaw);
    const uint32_t lane = threadIdx.x;
    const uint64_t* __restrict__ pairKeys = sc->partitioned ? pairsB : pairsA;
    for (uint32_t p = sc->gjkLo + blockIdx.x; p < sc->gjkHi; p += gridDim.x) {
        Shape sa, sb;
        const int mode = gjkPairShapes(pairKeys[p], wShape, sa, sb);
        if (mode < 0) continue;
        Simplex sx; Manifold m; m.count = 0;
        const int r = gjkPhaseWave(sa, sb, hs, mode, sx, m, lane);
        if (r == 2) epaPhaseWave(sa, sb, hs, mode, sx, lds, m, lane);
        if (lane == 0) {
            const uint32_t cnt = r ? m.count : 0u;
            npPacked[p] = cnt ? ((1ull << 32) | (uint64_t)cnt) : 0ull;
            if (cnt) {
                npNormal[p] = f4(m.n, 0.f);
                for (uint32_t k = 0; k < cnt; ++k) npPoints[4 * p + k] = f4(m.p[k], m.d[k]);
            }
        }
        __syncthreads();
    }
}
constexpr uint32_t kEpaSimplexRows = 9;   // 4 support points x (a, b, m) = 36 floats
__global__ __launch_bounds__(64) void k_narrow_gjk(StepScalars* __restrict__ sc, const uint64_t* __restrict__ pairsA, const uint64_t* __restrict__ pairsB,
                                                   const float4* __restrict__ wShape,
                                                   HullSet hs, uint64_t* __restrict__ npPacked, float4* __restrict__ npNormal,
                                                   float4* __restrict__ npPoints, uint32_t* __restrict__ epaQueue, float4* __restrict__ epaSimplex, uint32_t epaCap) {
    // [gjkLo, gjkHi) = span of the bucket-partitioned pair list that holds the GJK/EPA buckets (k_pair_finish)
    uint32_t p = sc->gjkLo + blockIdx.x * blockDim.x + threadIdx.x;
    int r = -1; Simplex sx; Manifold m; m.count = 0;
    if (p < sc->gjkHi) {
        const uint64_t* __restrict__ pairKeys = sc->partitioned ? pairsB : pairsA;
        Shape sa, sb;
        const int mode = gjkPairShapes(pairKeys[p], wShape, sa, sb);
        if (mode >= 0) r = gjkPhase(sa, sb, hs, mode, sx, m);
    }
    if (r == 0 || r == 1) {
        uint32_t cnt = r == 1 ? m.count : 0u;
        npPacked[p] = cnt ? ((1ull << 32) | (uint64_t)cnt) : 0ull;
        if (cnt) {
            npNormal[p] = f4(m.n, 0.f);
            for (uint32_t k = 0; k < cnt; ++k) npPoints[4 * p + k] = f4(m.p[k], m.d[k]);
        }
    }
    // queue slots of the intersecting pairs: one atomic per wave
    const bool want = r == 2;
    const unsigned long long mask = __ballot(want);
    if (!mask) return;
    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&sc->numEpa, (uint32_t)__popcll(mask));
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    if (!want) return;
    const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (slot >= epaCap) { sc->specOverflow = 1u; npPacked[p] = 0ull; return; }   // cannot happen: the queue holds as many entries as there are pairs
    epaQueue[slot] = p;
    float4* o = epaSimplex + (size_t)slot * kEpaSimplexRows;
    const SupPt* q[4] = {&sx.a, &sx.b, &sx.c, &sx.d};
    float f[36];
#pragma unroll
    for (int k = 0; k < 4; ++k) { f[9 * k] = q[k]->a.x; f[9 * k + 1] = q[k]->a.y; f[9 * k + 2] = q[k]->a.z; f[9 * k + 3] = q[k]->b.x; f[9 * k + 4] = q[k]->b.y; f[9 * k + 5] = q[k]->b.z; f[9 * k + 6] = q[k]->m.x; f[9 * k + 7] = q[k]->m.y; f[9 * k + 8] = q[k]->m.z; }
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
}
__global__ __launch_bounds__(64) void k_narrow_epa(const StepScalars* __restrict__ sc, const uint64_t* __restrict__ pairsA, const uint64_t* __restrict__ pairsB,
                                                   const float4* __restrict__ wShape, HullSet hs, const uint32_t* __restrict__ epaQueue, const float4* __restrict__ epaSimplex,
                                                   uint32_t epaCap, uint64_t* __restrict__ npPacked, float4* __restrict__ npNormal, float4* __restrict__ npPoints) {
    __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[sizeof(EpaLds)];   // (V3 has constructors: raw storage)
    EpaLds& lds = *reinterpret_cast<EpaLds*>(ldsRaw);
    const uint32_t lane = threadIdx.x;
    const uint32_t n = min(sc->numEpa, epaCap);
    const uint64_t* __restrict__ pairKeys = sc->partitioned ? pairsB : pairsA;
    for (uint32_t entry = blockIdx.x; entry < n; entry += gridDim.x) {
        const uint32_t p = epaQueue[entry];
        Shape sa, sb;
        const int mode = gjkPairShapes(pairKeys[p], wShape, sa, sb);
        const float4* q = epaSimplex + (size_t)entry * kEpaSimplexRows;
        float f[36];
#pragma unroll
        for (int k = 0; k < 9; ++k) { const float4 v = q[k]; f[4 * k] = v.x; f[4 * k + 1] = v.y; f[4 * k + 2] = v.z; f[4 * k + 3] = v.w; }
        Simplex sx; sx.n = 4;
        SupPt* d[4] = {&sx.a, &sx.b, &sx.c, &sx.d};
#pragma unroll
        for (int k = 0; k < 4; ++k) { d[k]->a = V3(f[9 * k], f[9 * k + 1], f[9 * k + 2]); d[k]->b = V3(f[9 * k + 3], f[9 * k + 4], f[9 * k + 5]); d[k]->m = V3(f[9 * k + 6], f[9 * k + 7], f[9 * k + 8]); }
        Manifold m; m.count = 0;
        epaPhaseWave(sa, sb, hs, mode, sx, lds, m, lane);
        if (lane == 0) {
            const uint32_t cnt = m.count;
            npPacked[p] = cnt ? ((1ull << 32) | (uint64_t)cnt) : 0ull;
            if (cnt) {
                npNormal[p] = f4(m.n, 0.f);
                for (uint32_t k = 0; k < cnt; ++k) npPoints[4 * p + k] = f4(m.p[k], m.d[k]);
            }
        }
        __syncthreads();   // the polytope in LDS is reused by the next entry
    }
}

}  // namespace mi
