// heightmap.hpp — heightmap terrain narrow phase on the device.
//
// heightmapCollision (src/physics/heightmap_collision.cpp:509-618) over the min/max-mip quadtree of
// src/terrain/heightmap_collider.h:35-118, 153-207.  Every rigid-body collider visits the quads of the chunks its
// (upward-extended) AABB touches and tests the surviving triangles; contacts are numbered in the reference's stack (LIFO)
// order, so contact j of a collider is the same contact as in the sequential code.  Every contact becomes a one-contact manifold
// appended to the pair list AFTER the collider-pair narrow phase: key = (bucket 21, collider, kHeightmapVirtualBase + j),
// body B = the static dummy.  From there on the contacts take the normal path (colouring, constraint init, solver).
//
// The usual collider spans a few terrain cells: one wave per collider tests those cells in parallel (k_hm_contacts, see the
// pipeline comment below); the sequential stack walk remains for colliders with a large cell window.
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

// The triangle test of one collider shape (the four `intersection` overloads, heightmap_collision.cpp:431-507).
struct TriShape {
    Shape s; V3 boxCenter, boxRadius, capDir; Q4 inv;
    __device__ explicit TriShape(const Shape& sh) : s(sh) {
        boxCenter = s.type == T_AABB ? (s.a + s.b) * 0.5f : V3();
        boxRadius = s.type == T_AABB ? (s.b - s.a) * 0.5f : s.b;
        inv = conj(s.rot);
        capDir = s.type == T_CAPSULE ? normalize(s.b - s.a) : V3();
    }
    __device__ bool test(V3 a, V3 b, V3 c, TriContact& t) const {
        if (s.type == T_CYLINDER || s.type == T_HULL) return false;   // no shape-vs-triangle routine in the reference: only the lowest point is tested (below)
        if (s.type == T_SPHERE) return sphereVsTriangle(s.a, s.radius, a, b, c, t);
        if (s.type == T_CAPSULE) {   // heightmap_collision.cpp:445-471
            V3 triNormal = normalize(cross(b - a, c - a));
            float d = -dot(triNormal, a);
            float ndotd = dot(capDir, triNormal);
            float tt = -(dot(s.a, triNormal) + d) / ndotd;
            V3 trace = s.a + tt * capDir;
            V3 closest = closestOnTriangle(trace, a, b, c);
            V3 reference = closestOnSegment(closest, s.a, s.b);
            return sphereVsTriangle(reference, s.radius, a, b, c, t);
        }
        // boxes: an OBB takes the triangle into its frame and the contact back (heightmap_collision.cpp:473-507)
        const bool obb = s.type == T_OBB;
        if (obb) { a = rotate(inv, a - s.a); b = rotate(inv, b - s.a); c = rotate(inv, c - s.a); }
        bool hit = boxVsTriangle(boxCenter, boxRadius, a, b, c, t);
        if (hit && obb) { t.normal = rotate(s.rot, t.normal); t.point = rotate(s.rot, t.point) + s.a; }
        return hit;
    }
};

// heightmap_collider_component::iterateTrianglesInVolume, outer level (heightmap_collider.h:153-205): the chunk range and,
// per chunk, the cell window [volMinX, volMaxX] x [volMinZ, volMaxZ] and the uint16 height window.
struct HmVolume {
    V3 vmin, vmax, corner;
    uint32_t minCX, minCZ, maxCX, maxCZ, volMinY, volMaxY;
    __device__ HmVolume(const HeightmapParams& hm, V3 mn, V3 mx) : corner(hm.minX, hm.minY, hm.minZ) {
        vmin = mn - corner; vmax = mx - corner;
        vmin.x *= hm.invChunkSize; vmin.z *= hm.invChunkSize; vmax.x *= hm.invChunkSize; vmax.z *= hm.invChunkSize;
        const int cpd = (int)hm.chunksPerDim;
        minCX = (uint32_t)max((int)vmin.x, 0); minCZ = (uint32_t)max((int)vmin.z, 0);
        maxCX = (uint32_t)min(max((int)vmax.x, 0), cpd - 1); maxCZ = (uint32_t)min(max((int)vmax.z, 0), cpd - 1);
        vmin.y *= hm.invAmplitudeScale; vmax.y *= hm.invAmplitudeScale;
        volMinY = hmToU32(clamp01(vmin.y) * 65535.f) & 0xFFFFu; volMaxY = hmToU32(clamp01(vmax.y) * 65535.f) & 0xFFFFu;
    }
    __device__ void window(uint32_t x, uint32_t z, uint32_t& x0, uint32_t& z0, uint32_t& x1, uint32_t& z1) const {
        const float relMinX = fmaxr(vmin.x - (float)x, 0.f), relMinZ = fmaxr(vmin.z - (float)z, 0.f);
        const float relMaxX = (vmax.x > (float)(x + 1u)) ? 1.f : hmFrac(vmax.x), relMaxZ = (vmax.z > (float)(z + 1u)) ? 1.f : hmFrac(vmax.z);
        x0 = hmToU32(relMinX * (float)kHmVerts); z0 = hmToU32(relMinZ * (float)kHmVerts);
        x1 = hmToU32(relMaxX * (float)kHmVerts); z1 = hmToU32(relMaxZ * (float)kHmVerts);
    }
};
__device__ __forceinline__ V3 hmVertex(const HeightmapParams& hm, const uint16_t* __restrict__ heights, V3 chunkMin, uint32_t vx, uint32_t vz) {
    float h = (float)heights[kHmVerts * vz + vx] * hm.heightScale;
    return V3((float)vx * hm.chunkScale, h, (float)vz * hm.chunkScale) + chunkMin;
}
// the collider's lowest point under the bilinear surface (heightmap_collision.cpp:572-580)
__device__ inline bool hmLowestPoint(const HeightmapParams& hm, const Shape& s, const HullSet& hulls, TriContact& t) {
    V3 lowest = supportOf(s, hulls, V3(0.f, -1.f, 0.f));
    float h = hmHeightAt(hm, lowest.x, lowest.z);
    if (!(lowest.y < h)) return false;
    t.point = lowest; t.normal = V3(0.f, -1.f, 0.f); t.depth = h - lowest.y;
    return true;
}

// ---- the device pipeline ------------------------------------------------------------------------------------------------
// k_hm_contacts<WRITE, LARGE>  one WAVE per collider, one lane per TRIANGLE of the collider's cell window.  The reference's stack walk
//                 (heightmap_collider.h:35-118: root of the mip pyramid on a stack; a node is dropped when it misses the window or the
//                 height range, a leaf tests its two triangles, an inner node pushes its children (0,0) (0,1) (1,0) (1,1)) visits exactly
//                 the quads that pass their own x/z and min/max-height test (an ancestor's box contains the quad's), in DESCENDING Morton
//                 order with x as the high bit, first triangle before second.  So the lanes test their triangles independently and
//                 contact j = the number of hit triangles that precede it in that order (found by comparing sort keys against the hit
//                 lanes only).  LARGE = false (every collider): windows of at most 64 cells per chunk — the usual case, a body spans a
//                 few cells; a collider with a larger window is flagged.  LARGE = true (launched behind it, flagged colliders only, the
//                 others' waves leave at once): the window's aligned 8 x 8 blocks (nodes of mip 3) in descending Morton order, pruned by
//                 their height range like the walk prunes them — the same order, 64 quads at a time (round 6; such colliders used to be
//                 walked by ONE lane in a kernel of their own).  WRITE = false: count per collider.  WRITE = true: the same again,
//                 contacts written straight to their final slots.
// (exclusive scan of the packed counts on the stream: contact offsets + colliders touching the terrain)
// k_hm_totals     StepScalars::numHmContacts / numHmColliders for the host's sizing read-back.
// k_hm_finish     numPairs += numHmContacts (after the WRITE passes, which address slots relative to the collider pairs).
// Counting happens with the world colliders (the host sizes the narrow-phase buffers from the totals), writing after the
// collider-pair narrow phase: slot = numPairs + offset(collider) + j — deterministic positions, no atomics.
__device__ __forceinline__ uint32_t hmSpread7(uint32_t v) {   // bit i -> bit 2 i (7 bits)
    v = (v | (v << 4)) & 0x0F0Fu; v = (v | (v << 2)) & 0x3333u; v = (v | (v << 1)) & 0x5555u;
    return v;
}
__device__ __forceinline__ uint32_t hmCompact4(uint32_t v) {   // bit 2 i -> bit i (4 bits)
    v &= 0x55u; v = (v | (v >> 1)) & 0x33u; v = (v | (v >> 2)) & 0x0Fu;
    return v;
}
__device__ __forceinline__ bool hmActive(uint32_t tag, uint32_t& type) {
    type = tag & 0xFFu;
    // cylinders and hulls too: the reference's switch has no case for them and reads an UNINITIALISED lowestPoint (heightmap_collision.cpp:533-573: undefined
    // behaviour); what it evidently means — their lowest point against the surface, like every other type, no triangle routine — is what runs here
    return ((tag >> 8) & 0xFFu) == OBJ_RIGID_BODY;
}
struct HmOut {   // where the WRITE passes put contact j of collider i
    StepScalars* sc; uint32_t pairCap; uint64_t* pairsA; uint64_t* pairsB; uint64_t* npPacked; float4* npNormal; float4* npPoints;
    __device__ bool ready() const { return !sc->specOverflow && sc->numPairs + sc->numHmContacts <= pairCap; }
    // Contact j of the collider's `count` goes to pair record first + j (one record per contact: point, depth, ITS normal).  Four consecutive records are ONE manifold
    // (round 6): the record of contact 4 g is the manifold's head (flag 1, contacts min(4, count - 4 g), key (collider, virtual index base + g)), the other three carry
    // (flag 0, 0 contacts) and only lend their point and normal — k_contact_init reads contact k of a terrain manifold from record head + k.  A box resting on eight
    // terrain contacts is two manifolds of one lane each, not eight colours.
    __device__ void put(uint32_t first, uint32_t i, uint32_t j, uint32_t count, const TriContact& t) const {
        const uint32_t p = first + j, g = j >> 2;
        (sc->partitioned ? pairsB : pairsA)[p] = ((uint64_t)kHmBucket << 58) | ((uint64_t)i << 29) | (uint64_t)(kHeightmapVirtualBase + g);
        npPacked[p] = (j & 3u) ? 0ull : ((1ull << 32) | (unsigned long long)min(4u, count - j));
        npNormal[p] = f4(t.normal, 0.f);
        npPoints[4 * (size_t)p] = f4(t.point, t.depth);
    }
};

// The counting pass keeps WHICH triangles it hit: the first kHmStash hits of a collider go to a stash ([collider][kHmStash] packed (chunk, quad, triangle); the
// lowest-point contact of a capsule / cylinder / hull as kHmStashLowest) and the WRITE pass recomputes just those — one lane per contact — instead of walking the
// collider's whole window of triangles a second time (65 536 bodies on terrain: the second walk was 212 us).  The counting pass itself stays what it was (it never
// needed the contact geometry: keeping the contacts themselves doubled its time).  A collider with more hits than the stash holds is walked again as before.
constexpr uint32_t kHmLowBit = 2u;          // hmSlow[i]: bit 0 = large window (set by the plain count pass), bit 1 = the lowest point is under the surface (k_hm_lowest, for the count passes)
constexpr uint32_t kHmStashLowest = 0xFFFFFFFFu;
constexpr uint32_t kHmStash = 16;
template <bool WRITE, bool LARGE>
__device__ __forceinline__ void hmCollider(const uint32_t i, const uint32_t lane, uint32_t nc, const HeightmapParams& hm, const float4* __restrict__ wShape, const float4* __restrict__ aabbMin,
                                           const float4* __restrict__ aabbMax, unsigned long long* __restrict__ hmPacked, uint8_t* __restrict__ hmSlow,
                                           const unsigned long long* __restrict__ hmScan, const HmOut& out, const HullSet& hulls, uint32_t* __restrict__ stash) {
    if (i >= nc) return;
    const float4 mn = aabbMin[i], mx = aabbMax[i];
    const bool low = !WRITE && (hmSlow[i] & kHmLowBit) != 0;                       // k_hm_lowest's answer for this collider (count passes)
    uint32_t type;
    const bool active = hmActive(__float_as_uint(mn.w), type) && !(mx.x < mn.x);   // (inverted box: sharded world, a body this rank does not simulate this step)
    uint32_t count = 0, first = 0;
    if (WRITE) {
        count = active ? (uint32_t)hmPacked[i] : 0u;
        if (!count || (hmSlow[i] != 0) != LARGE || !out.ready()) return;
        first = out.sc->numPairs + (uint32_t)hmScan[i];
        if (!LARGE && stash && count <= kHmStash && hm.chunksPerDim <= 256u) return;   // every hit of this collider is in the stash: k_hm_write_stashed recomputes just those
    } else if (LARGE) { if (!active || !hmSlow[i]) return; }                            // (the flags are the plain instance's, launched before this one)
    else if (!active) { if (lane == 0) { hmPacked[i] = 0ull; hmSlow[i] = 0; } return; }
    auto keep = [&](uint32_t j, uint32_t id) { if (!LARGE && stash && j < kHmStash) stash[(size_t)i * kHmStash + j] = id; };
    const Shape s = loadShape(wShape, i, type);
    const TriShape ts(s);
    const HmVolume vol(hm, xyz(mn), V3(mx.x, mx.y + 10.f, mx.z));
    uint32_t found = 0; bool slow = false;
    const bool triangles = type != T_CYLINDER && type != T_HULL;     // (wave-uniform: one collider per wave)
    for (uint32_t z = vol.minCZ; triangles && z <= vol.maxCZ && !slow; ++z)
        for (uint32_t x = vol.minCX; x <= vol.maxCX; ++x) {
            const uint32_t slot = hm.chunkSlot[z * hm.chunksPerDim + x];
            if (slot == 0xFFFFFFFFu) continue;
            uint32_t x0, z0, x1, z1;
            vol.window(x, z, x0, z0, x1, z1);
            x1 = min(x1, kHmSegs - 1u); z1 = min(z1, kHmSegs - 1u);
            if (x0 > x1 || z0 > z1) continue;
            const V3 chunkMin = V3((float)x * hm.chunkSize, 0.f, (float)z * hm.chunkSize) + vol.corner;
            const uint16_t* __restrict__ heights = hm.heights + (size_t)slot * kHmVerts * kHmVerts;
            // one batch: the quads of the rectangle [rx0, rx1] x [rz0, rz1] (at most 64 of them), in the reference's order (descending Morton code, first triangle first)
            auto batch = [&](const uint32_t x0, const uint32_t z0, const uint32_t x1, const uint32_t z1) {
            const uint32_t w = x1 - x0 + 1u, n = w * (z1 - z0 + 1u);
            // item = 2 * quad + triangle, 64 items per batch, at most two batches
            uint32_t sk[2] = {0u, 0u}; bool hit[2] = {false, false}; TriContact tc[2];
#pragma unroll
            for (uint32_t b = 0; b < 2u; ++b) {
                const uint32_t item = b * 64u + lane;
                if (b * 64u >= 2u * n) break;          // wave-uniform
                if (item < 2u * n) {
                    const uint32_t q = item >> 1, tri = item & 1u, qx = x0 + q % w, qz = z0 + q / w;
                    sk[b] = (((hmSpread7(qx) << 1) | hmSpread7(qz)) << 1) | (1u - tri);   // descending: larger Morton first, first triangle first
                    const uint32_t ha = heights[kHmVerts * qz + qx], hb = heights[kHmVerts * (qz + 1u) + qx], hc = heights[kHmVerts * qz + qx + 1u], hd = heights[kHmVerts * (qz + 1u) + qx + 1u];
                    const uint32_t lo = min(min(ha, hb), min(hc, hd)), hi = max(max(ha, hb), max(hc, hd));
                    if (!(hi < vol.volMinY || lo > vol.volMaxY)) {
                        // triangles (A, B, C) and (C, B, D) of the quad (heightmap_collider.h:85-106)
                        const V3 pb = hmVertex(hm, heights, chunkMin, qx, qz + 1u), pc = hmVertex(hm, heights, chunkMin, qx + 1u, qz);
                        const V3 pe = tri ? hmVertex(hm, heights, chunkMin, qx + 1u, qz + 1u) : hmVertex(hm, heights, chunkMin, qx, qz);
                        hit[b] = tri ? ts.test(pc, pb, pe, tc[b]) : ts.test(pe, pb, pc, tc[b]);
                    }
                }
            }
            const unsigned long long m0 = __ballot(hit[0]), m1 = __ballot(hit[1]);
            uint32_t before[2] = {0u, 0u};
            for (unsigned long long m = m0; m; m &= m - 1ull) {
                const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)sk[0], (int)(__ffsll((long long)m) - 1));
                before[0] += k > sk[0] ? 1u : 0u; before[1] += k > sk[1] ? 1u : 0u;
            }
            for (unsigned long long m = m1; m; m &= m - 1ull) {
                const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)sk[1], (int)(__ffsll((long long)m) - 1));
                before[0] += k > sk[0] ? 1u : 0u; before[1] += k > sk[1] ? 1u : 0u;
            }
            if (WRITE) {
#pragma unroll
                for (uint32_t b = 0; b < 2u; ++b) if (hit[b] && found + before[b] < count) out.put(first, i, found + before[b], count, tc[b]);
            } else {
#pragma unroll
                for (uint32_t b = 0; b < 2u; ++b) if (hit[b]) {
                    const uint32_t item = b * 64u + lane, q = item >> 1;
                    keep(found + before[b], (x << 23) | (z << 15) | ((x0 + q % w) << 8) | ((z0 + q / w) << 1) | (item & 1u));
                }
            }
            found = min(found + (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1), kHmMaxContacts);
            };
            // A window of at most 64 cells is one batch.  A larger one flags the collider (plain instance) and is taken by the LARGE instance of this kernel, launched right
            // behind it over the flagged colliders only (round 6; one lane used to walk these alone in k_hm_slow — 35 ms for a collider spanning a chunk of fine terrain, found
            // by tools/gpu_fuzz.py): the window's aligned 8 x 8 blocks in descending Morton order — which IS the order of the reference's walk down the mip pyramid
            // (heightmap_collider.h:35-118), a block being one node of mip 3 —, a block whose height range misses the collider's skipped as the walk prunes it.  (Two instances:
            // the block loop costs the plain one, which every collider of a terrain scene runs, ten registers and with them a wave of occupancy.)
            const bool whole = (x1 - x0 + 1u) * (z1 - z0 + 1u) <= 64u;
            if (!LARGE) {
                if (!whole) { slow = true; break; }
                batch(x0, z0, x1, z1);
            } else {
                const uint32_t* __restrict__ mip3 = hm.mips + (size_t)slot * kHmMipEntries + hmMipOffset(3u);
                for (int code = whole ? 0 : 255; code >= 0; --code) {
                    uint32_t rx0 = x0, rz0 = z0, rx1 = x1, rz1 = z1;
                    if (!whole) {
                        const uint32_t bx = hmCompact4((uint32_t)code >> 1), bz = hmCompact4((uint32_t)code);
                        if (bx < (x0 >> 3) || bx > (x1 >> 3) || bz < (z0 >> 3) || bz > (z1 >> 3)) continue;
                        const uint32_t mm = mip3[bz * (kHmSegs >> 3) + bx];
                        if ((mm >> 16) < vol.volMinY || (mm & 0xFFFFu) > vol.volMaxY) continue;
                        rx0 = max(x0, bx << 3); rz0 = max(z0, bz << 3); rx1 = min(x1, (bx << 3) + 7u); rz1 = min(z1, (bz << 3) + 7u);
                    }
                    batch(rx0, rz0, rx1, rz1);
                }
            }
        }
    if (lane != 0) return;
    if (WRITE) {
        TriContact t;
        if (found < count && hmLowestPoint(hm, s, hulls, t)) out.put(first, i, found, count, t);
        return;
    }
    if (slow) { hmPacked[i] = 0ull; hmSlow[i] = (uint8_t)(1u | (low ? kHmLowBit : 0u)); return; }
    if (low && found < kHmMaxContacts) { keep(found, kHmStashLowest); ++found; }
    hmPacked[i] = (unsigned long long)found | (found ? 1ull << 32 : 0ull);
    if (!LARGE) hmSlow[i] = 0;
}
// The lowest-point test of every collider (heightmap_collision.cpp:572-580), one LANE per collider, ahead of the count passes: its support point and its two dependent round
// trips (chunk slot, four heights) used to sit at the end of every wave of the count pass, on one lane.
__global__ __launch_bounds__(256) void k_hm_lowest(uint32_t nc, HeightmapParams hm, const float4* __restrict__ wShape, const float4* __restrict__ aabbMin, const float4* __restrict__ aabbMax,
                                                   uint8_t* __restrict__ hmSlow, HullSet hulls) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc) return;
    const float4 mn = aabbMin[i], mx = aabbMax[i];
    uint32_t type, r = 0;
    if (hmActive(__float_as_uint(mn.w), type) && !(mx.x < mn.x)) {
        const Shape s = loadShape(wShape, i, type);
        TriContact t;
        if (hmLowestPoint(hm, s, hulls, t)) r = kHmLowBit;
    }
    hmSlow[i] = (uint8_t)r;
}
constexpr uint32_t kHmScanBlocks = 512;   // the flag-scanning launches (LARGE or WRITE): at most this many workgroups, whatever the collider count
template <bool WRITE, bool LARGE>
__global__ __launch_bounds__(256) void k_hm_contacts(uint32_t nc, HeightmapParams hm, const float4* __restrict__ wShape, const float4* __restrict__ aabbMin,
                                                     const float4* __restrict__ aabbMax, unsigned long long* __restrict__ hmPacked, uint8_t* __restrict__ hmSlow,
                                                     const unsigned long long* __restrict__ hmScan, HmOut out, HullSet hulls, uint32_t* __restrict__ stash) {
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (!WRITE && !LARGE) { hmCollider<false, false>(wave, lane, nc, hm, wShape, aabbMin, aabbMax, hmPacked, hmSlow, hmScan, out, hulls, stash); return; }
    // Every other instance has work for a few colliders only (the flagged ones; in the WRITE passes those with more hits than the stash holds): a wave reads what decides
    // that for 64 colliders at a time and takes the ones that are its business one after the other.  (As launches of a wave per collider, all but a few of which left at once,
    // these cost the 65 536-body terrain scene 8-35 us each: three waves of 216-243 registers per SIMD, sixteen thousand workgroups.)
    for (uint32_t base = wave * 64u; base < nc; base += gridDim.x * 256u) {
        const uint32_t k = base + lane;
        bool mine = k < nc && (hmSlow[k] != 0) == LARGE;
        if (WRITE && mine) {
            const uint32_t count = (uint32_t)hmPacked[k];
            mine = count && (LARGE || !(stash && count <= kHmStash && hm.chunksPerDim <= 256u));
        }
        for (unsigned long long m = __ballot(mine); m; m &= m - 1ull)
            hmCollider<WRITE, LARGE>(base + (uint32_t)__ffsll((long long)m) - 1u, lane, nc, hm, wShape, aabbMin, aabbMax, hmPacked, hmSlow, hmScan, out, hulls, stash);
    }
}
// WRITE pass for the stashed colliders: one LANE per terrain contact.  Contact t belongs to the collider i with offset(i) <= t < offset(i) + count(i) (binary search
// over the scanned counts) and is its hit number j = t - offset(i): the lane recomputes that one triangle (or the lowest point) and writes the contact to its final slot.
__global__ __launch_bounds__(256) void k_hm_write_stashed(uint32_t nc, HeightmapParams hm, const float4* __restrict__ wShape, const float4* __restrict__ aabbMin,
                                                          const float4* __restrict__ aabbMax, const unsigned long long* __restrict__ hmPacked, const uint8_t* __restrict__ hmSlow,
                                                          const unsigned long long* __restrict__ hmScan, HmOut out, HullSet hulls, const uint32_t* __restrict__ stash) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= out.sc->numHmContacts || !out.ready() || hm.chunksPerDim > 256u) return;
    uint32_t lo = 0, hi = nc;                       // the last collider whose offset is <= t (colliders without contacts share their successor's offset)
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)hmScan[mid] <= t) lo = mid; else hi = mid; }
    const uint32_t i = lo, count = (uint32_t)hmPacked[i], j = t - (uint32_t)hmScan[i];
    if (j >= count || count > kHmStash || hmSlow[i]) return;     // (the wave-per-collider passes write those)
    const float4 mn = aabbMin[i], mx = aabbMax[i];
    const Shape s = loadShape(wShape, i, __float_as_uint(mn.w) & 0xFFu);
    const uint32_t id = stash[(size_t)i * kHmStash + j];
    TriContact tc; bool ok;
    if (id == kHmStashLowest) ok = hmLowestPoint(hm, s, hulls, tc);
    else {
        const uint32_t x = id >> 23, z = (id >> 15) & 0xFFu, qx = (id >> 8) & 0x7Fu, qz = (id >> 1) & 0x7Fu, tri = id & 1u;
        const HmVolume vol(hm, xyz(mn), V3(mx.x, mx.y + 10.f, mx.z));
        const V3 chunkMin = V3((float)x * hm.chunkSize, 0.f, (float)z * hm.chunkSize) + vol.corner;
        const uint16_t* __restrict__ heights = hm.heights + (size_t)hm.chunkSlot[z * hm.chunksPerDim + x] * kHmVerts * kHmVerts;
        const TriShape ts(s);
        const V3 pb = hmVertex(hm, heights, chunkMin, qx, qz + 1u), pc = hmVertex(hm, heights, chunkMin, qx + 1u, qz);
        const V3 pe = tri ? hmVertex(hm, heights, chunkMin, qx + 1u, qz + 1u) : hmVertex(hm, heights, chunkMin, qx, qz);
        ok = tri ? ts.test(pc, pb, pe, tc) : ts.test(pe, pb, pc, tc);
    }
    if (ok) out.put(out.sc->numPairs + (uint32_t)hmScan[i], i, j, count, tc);
}
__global__ void k_hm_totals(uint32_t nc, const unsigned long long* __restrict__ hmPacked, const unsigned long long* __restrict__ hmScan, StepScalars* sc) {
    const unsigned long long t = nc ? hmScan[nc - 1u] + hmPacked[nc - 1u] : 0ull;
    sc->numHmContacts = (uint32_t)t; sc->numHmColliders = (uint32_t)(t >> 32);
}
__global__ void k_hm_finish(StepScalars* sc, uint32_t pairCap) {
    if (sc->specOverflow) return;
    if (sc->numPairs + sc->numHmContacts > pairCap) { sc->specOverflow = 1u; return; }
    sc->numPairs += sc->numHmContacts;
}

}  // namespace mi
