// heightmap.hpp — heightmap terrain narrow phase on the device.
//
// heightmapCollision (src/physics/heightmap_collision.cpp:509-618) over the min/max-mip quadtree of
// src/terrain/heightmap_collider.h:35-118, 153-207.  One lane per rigid-body collider walks the quadtree of the chunks its
// (upward-extended) AABB touches and tests the surviving triangles; the triangle order is the reference's stack (LIFO) order,
// so contact j of a collider is the same contact as in the sequential code.  Every contact becomes a one-contact manifold
// appended to the pair list AFTER the collider-pair narrow phase: key = (bucket 21, collider, kHeightmapVirtualBase + j),
// body B = the static dummy.  From there on the contacts take the normal path (colouring, constraint init, solver).
//
// Two passes over the same walk: k_heightmap<false> counts (per collider and in total, so the host / the speculative bounds
// can size the buffers), k_heightmap<true> reserves one slot range per workgroup and writes.
//
// Stated deviations from the reference: cylinder / hull colliders are skipped (the reference reads an uninitialised point
// for them); float -> uint32 conversions of possibly negative values go through int64 (what x86-64 code does); at most 255
// contacts per collider (the reference asserts it).
#pragma once
#include "kernels.hpp"
#include "gjk.hpp"

namespace mi {

constexpr uint32_t kHmVerts = 129u, kHmSegs = 128u;          // TERRAIN_LOD_0_VERTICES_PER_DIMENSION
constexpr uint32_t kHmMipEntries = 21845u;                   // 128^2 + 64^2 + ... + 1
constexpr uint32_t kHmBucket = 21u;                          // pair-key bucket of heightmap contacts (collider pairs use 0..20)
constexpr uint32_t kHmMaxContacts = 255u;

struct HeightmapParams {
    const uint16_t* heights;      // [chunk slot][129 * 129]
    const uint32_t* mips;         // [chunk slot][kHmMipEntries]: min | max << 16, mip 0 first
    const uint32_t* chunkSlot;    // [chunksPerDim^2]: slot of the chunk's data or ~0u (no heights: collides with nothing)
    uint32_t chunksPerDim;
    float chunkSize, invChunkSize, chunkScale, heightScale, invAmplitudeScale;
    float minX, minY, minZ;
    float restitution, friction;
};

__host__ __device__ __forceinline__ uint32_t hmMipOffset(uint32_t mip) {   // entries before mip level `mip`
    uint32_t off = 0;
    for (uint32_t m = 0; m < mip; ++m) off += (kHmSegs >> m) * (kHmSegs >> m);
    return off;
}
__host__ __device__ __forceinline__ uint32_t hmToU32(float f) { return (uint32_t)(long long)f; }
__host__ __device__ __forceinline__ float hmFrac(float v) { return fmodf(v, 1.f); }   // math.h:40

// heightmap_collider_component::getHeightAt -> heightmap_collider_chunk::getHeightAt (heightmap_collider.cpp:21-40, 116-153)
__host__ __device__ inline float hmHeightAt(const HeightmapParams& hm, float wx, float wz) {
    float cx = (wx - hm.minX) * hm.invChunkSize, cz = (wz - hm.minZ) * hm.invChunkSize;
    if (cx < 0.f || cz < 0.f || cx >= (float)hm.chunksPerDim || cz >= (float)hm.chunksPerDim) return -FLT_MAX;
    uint32_t slot = hm.chunkSlot[hmToU32(cz) * hm.chunksPerDim + hmToU32(cx)];
    if (slot == 0xFFFFFFFFu) return -FLT_MAX;
    const uint16_t* h = hm.heights + (size_t)slot * kHmVerts * kHmVerts;
    float fx = hmFrac(cx) * (float)kHmSegs, fz = hmFrac(cz) * (float)kHmSegs;
    uint32_t x = hmToU32(fx), z = hmToU32(fz);
    float relX = fx - (float)x, relZ = fz - (float)z;
    float a = (float)h[kHmVerts * z + x] * hm.heightScale, b = (float)h[kHmVerts * (z + 1u) + x] * hm.heightScale;
    float c = (float)h[kHmVerts * z + x + 1u] * hm.heightScale, d = (float)h[kHmVerts * (z + 1u) + x + 1u] * hm.heightScale;
    return lerpr(lerpr(a, c, relX), lerpr(b, d, relX), relZ) + hm.minY;
}

// closestPoint_PointTriangle — bounding_volumes.cpp:1317-1367
__device__ inline V3 closestOnTriangle(V3 p, V3 a, V3 b, V3 c) {
    V3 ab = b - a, ac = c - a, ap = p - a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) return a;
    V3 bp = p - b;
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) return b;
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { float v = d1 / (d1 - d3); return a + v * ab; }
    V3 cp = p - c;
    float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) return c;
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { float w = d2 / (d2 - d6); return a + w * ac; }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) { float w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); return b + w * (c - b); }
    float denom = 1.f / (va + vb + vc);
    float v = vb * denom, w = vc * denom;
    return a + ab * v + ac * w;
}

struct TriContact { V3 point, normal; float depth; };

// collideSphereVsTriangle — heightmap_collision.cpp:42-76
__device__ inline bool sphereVsTriangle(V3 center, float radius, V3 a, V3 b, V3 c, TriContact& out) {
    V3 closest = closestOnTriangle(center, a, b, c);
    V3 n = closest - center;
    float sq = sqlen(n);
    if (!(sq <= radius * radius)) return false;
    float distance;
    if (sq == 0.f) { n = -cross(b - a, c - a); distance = 0.f; }
    else { distance = sqrtf(sq); n = n * (1.f / distance); }
    out.point = closest; out.normal = n; out.depth = radius - distance;
    return true;
}

// collideAABBvsTriangle — heightmap_collision.cpp:78-429: 9 edge axes, 6 box-face directions, the triangle plane;
// minimum-penetration axis -> one contact (edge-edge closest points / deepest triangle vertex / box corner).
__device__ inline bool boxVsTriangle(V3 center, V3 radius, V3 a, V3 b, V3 c, TriContact& out) {
    a = a - center; b = b - center; c = c - center;
    float minPen = FLT_MAX; V3 minNormal; int category = 0;   // 0..2 edge axis of triangle edge k, 3 box face, 4 triangle plane
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const V3 e = (k == 0) ? b - a : (k == 1) ? c - b : a - c;
            const V3 q = (k == 0) ? c : b;
            float p0, p1, r; V3 n;
            if (axis == 0) { p0 = (a.z * e.y) - (a.y * e.z); p1 = (q.z * e.y) - (q.y * e.z); r = radius.y * fabsf(e.z) + radius.z * fabsf(e.y); n = V3(0.f, -e.z, e.y); }
            else if (axis == 1) { p0 = (a.x * e.z) - (a.z * e.x); p1 = (q.x * e.z) - (q.z * e.x); r = radius.x * fabsf(e.z) + radius.z * fabsf(e.x); n = V3(e.z, 0.f, -e.x); }
            else { p0 = (a.y * e.x) - (a.x * e.y); p1 = (q.y * e.x) - (q.x * e.y); r = radius.x * fabsf(e.y) + radius.y * fabsf(e.x); n = V3(-e.y, e.x, 0.f); }
            float pen = r - fmaxr(-fmaxr(p0, p1), fminr(p0, p1));
            if (pen < 0.f) return false;
            float l = len(n);
            pen *= 1.f / l;
            if (pen < minPen) { minPen = pen; minNormal = n * (1.f / l); category = k; }
        }
    }
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
        float av = a.get(axis), bv = b.get(axis), cv = c.get(axis), rv = radius.get(axis);
        float pen = fmaxr(av, fmaxr(bv, cv)) + rv;
        if (pen < 0.f) return false;
        if (pen < minPen) { minPen = pen; minNormal = V3(); minNormal.set(axis, -1.f); category = 3; }
        pen = rv - fminr(av, fminr(bv, cv));
        if (pen < 0.f) return false;
        if (pen < minPen) { minPen = pen; minNormal = V3(); minNormal.set(axis, 1.f); category = 3; }
    }
    {
        V3 triNormal = normalize(cross(b - a, c - b));
        float pen = dot(radius, vabs(triNormal)) - fabsf(dot(triNormal, a));
        if (pen < 0.f) return false;
        if (pen < minPen) { minPen = pen; minNormal = triNormal; category = 4; }
    }
    V3 triCenter = (a + b + c) * (1.f / 3.f);
    if (dot(minNormal, triCenter) < 0.f) minNormal = -minNormal;
    V3 point;
    if (category < 3) {
        // getAABBIncidentEdge — heightmap_collision.cpp:6-40
        V3 p = vabs(minNormal), e0 = radius, e1;
        if (p.x > p.y) e1 = (p.y > p.z) ? V3(radius.x, radius.y, -radius.z) : V3(radius.x, -radius.y, radius.z);
        else e1 = (p.x > p.z) ? V3(radius.x, radius.y, -radius.z) : V3(-radius.x, radius.y, radius.z);
        V3 s(minNormal.x < 0.f ? -1.f : 1.f, minNormal.y < 0.f ? -1.f : 1.f, minNormal.z < 0.f ? -1.f : 1.f);
        e0 = e0 * s; e1 = e1 * s;
        V3 ta = (category == 0) ? a : (category == 1) ? b : c;
        V3 tb = (category == 0) ? b : (category == 1) ? c : a;
        V3 pa, pb;
        closestSegmentSegment(e0, e1, ta, tb, pa, pb);
        point = (pa + pb) * 0.5f;
    } else if (category == 3) {
        float da = dot(minNormal, a), db = dot(minNormal, b), dc = dot(minNormal, c);
        V3 p = (da < db) ? ((da < dc) ? a : c) : ((db < dc) ? b : c);
        point = p + minNormal * (minPen * 0.5f);
    } else {
        V3 p(minNormal.x < 0.f ? -radius.x : radius.x, minNormal.y < 0.f ? -radius.y : radius.y, minNormal.z < 0.f ? -radius.z : radius.z);
        point = p - minNormal * (minPen * 0.5f);
    }
    out.point = point + center; out.normal = minNormal; out.depth = minPen;
    return true;
}

// One collider against the terrain.  `sink(j, contact)` receives contact j (j < 255) in the reference's order; returns the count.
template <typename Sink>
__device__ inline uint32_t heightmapContacts(const HeightmapParams& hm, const Shape& s, V3 vmin, V3 vmax, const Sink& sink) {
    uint32_t found = 0;
    const V3 corner(hm.minX, hm.minY, hm.minZ);
    // per-shape constants of the triangle tests
    const V3 boxCenter = s.type == T_AABB ? (s.a + s.b) * 0.5f : V3();
    const V3 boxRadius = s.type == T_AABB ? (s.b - s.a) * 0.5f : s.b;
    const Q4 inv = conj(s.rot);
    const V3 capDir = s.type == T_CAPSULE ? normalize(s.b - s.a) : V3();
    auto triangle = [&](V3 a, V3 b, V3 c) {
        TriContact t; bool hit;
        if (s.type == T_SPHERE) hit = sphereVsTriangle(s.a, s.radius, a, b, c, t);
        else if (s.type == T_CAPSULE) {   // heightmap_collision.cpp:445-471
            V3 triNormal = normalize(cross(b - a, c - a));
            float d = -dot(triNormal, a);
            float ndotd = dot(capDir, triNormal);
            float tt = -(dot(s.a, triNormal) + d) / ndotd;
            V3 trace = s.a + tt * capDir;
            V3 closest = closestOnTriangle(trace, a, b, c);
            V3 reference = closestOnSegment(closest, s.a, s.b);
            hit = sphereVsTriangle(reference, s.radius, a, b, c, t);
        } else if (s.type == T_AABB) hit = boxVsTriangle(boxCenter, boxRadius, a, b, c, t);
        else {                            // OBB: triangle into the box frame, contact back (heightmap_collision.cpp:492-507)
            hit = boxVsTriangle(V3(), boxRadius, rotate(inv, a - s.a), rotate(inv, b - s.a), rotate(inv, c - s.a), t);
            if (hit) { t.normal = rotate(s.rot, t.normal); t.point = rotate(s.rot, t.point) + s.a; }
        }
        if (hit && found < kHmMaxContacts) { sink(found, t); ++found; }
    };
    // heightmap_collider_component::iterateTrianglesInVolume — heightmap_collider.h:153-205
    vmin = vmin - corner; vmax = vmax - corner;
    vmin.x *= hm.invChunkSize; vmin.z *= hm.invChunkSize; vmax.x *= hm.invChunkSize; vmax.z *= hm.invChunkSize;
    const int cpd = (int)hm.chunksPerDim;
    const uint32_t minCX = (uint32_t)max((int)vmin.x, 0), minCZ = (uint32_t)max((int)vmin.z, 0);
    const uint32_t maxCX = (uint32_t)min(max((int)vmax.x, 0), cpd - 1), maxCZ = (uint32_t)min(max((int)vmax.z, 0), cpd - 1);
    vmin.y *= hm.invAmplitudeScale; vmax.y *= hm.invAmplitudeScale;
    const uint32_t volMinY = hmToU32(clamp01(vmin.y) * 65535.f) & 0xFFFFu, volMaxY = hmToU32(clamp01(vmax.y) * 65535.f) & 0xFFFFu;
    for (uint32_t z = minCZ; z <= maxCZ; ++z)
        for (uint32_t x = minCX; x <= maxCX; ++x) {
            const uint32_t slot = hm.chunkSlot[z * hm.chunksPerDim + x];
            if (slot == 0xFFFFFFFFu) continue;
            const float relMinX = fmaxr(vmin.x - (float)x, 0.f), relMinZ = fmaxr(vmin.z - (float)z, 0.f);
            const float relMaxX = (vmax.x > (float)(x + 1u)) ? 1.f : hmFrac(vmax.x), relMaxZ = (vmax.z > (float)(z + 1u)) ? 1.f : hmFrac(vmax.z);
            const uint32_t volMinX = hmToU32(relMinX * (float)kHmVerts), volMinZ = hmToU32(relMinZ * (float)kHmVerts);
            const uint32_t volMaxX = hmToU32(relMaxX * (float)kHmVerts), volMaxZ = hmToU32(relMaxZ * (float)kHmVerts);
            const V3 chunkMin = V3((float)x * hm.chunkSize, 0.f, (float)z * hm.chunkSize) + corner;
            const uint16_t* __restrict__ heights = hm.heights + (size_t)slot * kHmVerts * kHmVerts;
            const uint32_t* __restrict__ mips = hm.mips + (size_t)slot * kHmMipEntries;
            // heightmap_collider_chunk::iterateTrianglesInVolume — heightmap_collider.h:35-118.  Node = mip << 16 | x << 8 | z.
            uint32_t stack[28]; uint32_t top = 0;
            stack[top++] = 7u << 16;
            while (top) {
                const uint32_t e = stack[--top];
                const uint32_t mip = e >> 16, ex = (e >> 8) & 0xFFu, ez = e & 0xFFu;
                const uint32_t x0 = ex << mip, z0 = ez << mip, x1 = ((ex + 1u) << mip) - 1u, z1 = ((ez + 1u) << mip) - 1u;
                if (x1 < volMinX || x0 > volMaxX) continue;
                if (z1 < volMinZ || z0 > volMaxZ) continue;
                const uint32_t mm = mips[hmMipOffset(mip) + ez * (kHmSegs >> mip) + ex];
                if ((mm >> 16) < volMinY || (mm & 0xFFFFu) > volMaxY) continue;
                if (mip == 0u) {
                    auto vertex = [&](uint32_t vx, uint32_t vz) {
                        float h = (float)heights[kHmVerts * vz + vx] * hm.heightScale;
                        return V3((float)vx * hm.chunkScale, h, (float)vz * hm.chunkScale) + chunkMin;
                    };
                    V3 pa = vertex(ex, ez), pb = vertex(ex, ez + 1u), pc = vertex(ex + 1u, ez), pd = vertex(ex + 1u, ez + 1u);
                    triangle(pa, pb, pc);
                    triangle(pc, pb, pd);
                } else {
                    const uint32_t m = (mip - 1u) << 16;
                    stack[top++] = m | ((2u * ex) << 8) | (2u * ez);
                    stack[top++] = m | ((2u * ex) << 8) | (2u * ez + 1u);
                    stack[top++] = m | ((2u * ex + 1u) << 8) | (2u * ez);
                    stack[top++] = m | ((2u * ex + 1u) << 8) | (2u * ez + 1u);
                }
            }
        }
    // the collider's lowest point under the bilinear surface (heightmap_collision.cpp:572-580)
    HullSet none{nullptr, nullptr};
    V3 lowest = supportOf(s, none, V3(0.f, -1.f, 0.f));
    float h = hmHeightAt(hm, lowest.x, lowest.z);
    if (lowest.y < h && found < kHmMaxContacts) { TriContact t; t.point = lowest; t.normal = V3(0.f, -1.f, 0.f); t.depth = h - lowest.y; sink(found, t); ++found; }
    return found;
}

// WRITE = false: hmCount[collider] and the step totals.  WRITE = true: one slot range per workgroup at the end of the pair
// list (sc->numPairs grows), one-contact manifolds written straight into the narrow-phase output arrays.
template <bool WRITE>
__global__ __launch_bounds__(256) void k_heightmap(uint32_t nc, HeightmapParams hm, const float4* __restrict__ wShape, const float4* __restrict__ aabbMin,
                                                   const float4* __restrict__ aabbMax, uint32_t* __restrict__ hmCount, StepScalars* sc,
                                                   uint32_t pairCap, uint64_t* __restrict__ pairsA, uint64_t* __restrict__ pairsB,
                                                   uint64_t* __restrict__ npPacked, float4* __restrict__ npNormal, float4* __restrict__ npPoints) {
    __shared__ uint32_t waveSum[4], blockBase, blockColliders;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t type = 0; bool active = false;
    float4 mn, mx;
    if (i < nc) {
        mn = aabbMin[i]; mx = aabbMax[i];
        const uint32_t tag = __float_as_uint(mn.w);
        type = tag & 0xFFu;
        active = ((tag >> 8) & 0xFFu) == OBJ_RIGID_BODY && (type == T_SPHERE || type == T_CAPSULE || type == T_AABB || type == T_OBB);
    }
    uint32_t count = 0;
    if (WRITE) { if (active) count = hmCount[i]; if (sc->specOverflow) count = 0; }
    else if (active) {
        Shape s = loadShape(wShape, i, type);
        count = heightmapContacts(hm, s, xyz(mn), V3(mx.x, mx.y + 10.f, mx.z), [](uint32_t, const TriContact&) {});
        hmCount[i] = count;
    }
    // workgroup totals: inclusive wave scan, then one atomic per workgroup
    uint32_t incl = count;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t v = __shfl_up(incl, d); if ((int)lane >= d) incl += v; }
    if (lane == 63u) waveSum[wave] = incl;
    unsigned long long touching = __ballot(count != 0u);
    if (threadIdx.x == 0) blockColliders = 0;
    __syncthreads();
    if (lane == 0 && touching) atomicAdd(&blockColliders, (uint32_t)__popcll(touching));
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < 4; ++w) { if (w < wave) before += waveSum[w]; total += waveSum[w]; }
    __syncthreads();
    if (threadIdx.x == 0 && total) {
        if (WRITE) {
            uint32_t base = atomicAdd(&sc->numPairs, total);
            if (base + total > pairCap) { sc->specOverflow = 1u; base = 0xFFFFFFFFu; }
            blockBase = base;
        } else { atomicAdd(&sc->numHmContacts, total); atomicAdd(&sc->numHmColliders, blockColliders); }
    }
    if (!WRITE) return;
    __syncthreads();
    if (!count || blockBase == 0xFFFFFFFFu) return;
    const uint32_t first = blockBase + before + (incl - count);
    uint64_t* __restrict__ pairKeys = sc->partitioned ? pairsB : pairsA;
    Shape s = loadShape(wShape, i, type);
    heightmapContacts(hm, s, xyz(mn), V3(mx.x, mx.y + 10.f, mx.z), [&](uint32_t j, const TriContact& t) {
        if (j >= count) return;
        const uint32_t p = first + j;
        pairKeys[p] = ((uint64_t)kHmBucket << 58) | ((uint64_t)i << 29) | (uint64_t)(kHeightmapVirtualBase + j);
        npPacked[p] = (1ull << 32) | 1ull;
        npNormal[p] = f4(t.normal, 0.f);
        npPoints[4 * (size_t)p] = f4(t.point, t.depth);
    });
}

}  // namespace mi
