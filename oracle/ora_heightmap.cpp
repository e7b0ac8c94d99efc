// ora_heightmap.cpp — TEST INFRASTRUCTURE (CPU oracle), see ora_math.h header note.
//
// Heightmap terrain narrow phase: heightmapCollision (src/physics/heightmap_collision.cpp:509-618) over the min/max-mip
// quadtree of src/terrain/heightmap_collider.h:35-118, 153-207 and heightmap_collider.cpp:5-153.  Parity status: unpinned
// against the original binary (see ora_world.h); pinned by analytic known-answer tests (tests/test_oracle_kat.py).
//
// Stated deviations from the reference as written:
//  * cylinder and hull colliders are skipped: the reference's switch has no case for them and then reads an uninitialised
//    `lowestPoint` (heightmap_collision.cpp:533-570) — undefined behaviour, nothing to reproduce;
//  * float -> uint32 conversions of possibly negative values (frac() of a negative coordinate, heightmap_collider.h:187-194)
//    go through int64 like x86-64 code does, so they are defined and identical on the device;
//  * a collider reports at most 255 contacts (the reference asserts numContacts < 256, heightmap_collision.cpp:591);
//  * four consecutive heightmap contacts of a collider form one manifold {collider, kHeightmapVirtualBase + j / 4} (round 6; each contact keeps its own normal —
//    the solver works per contact anyway — so the sequence of updates in the reference's order is unchanged; in the canonical order a box on eight terrain
//    contacts is two manifolds, not eight colours); counts.num_collisions still counts one collision per collider like the reference.
#include "ora_world.h"
#include <algorithm>

namespace ora {

static const uint32_t kVerts = 129u;   // TERRAIN_LOD_0_VERTICES_PER_DIMENSION (heightmap_collider.h:8-11)
static const uint32_t kSegs = 128u;

// heightmap_collider_chunk::setHeights — heightmap_collider.cpp:42-114: min/max pyramid, mip 0 = one entry per quad
void Heightmap::setHeights(uint32_t x, uint32_t z, const uint16_t* h) {
    Chunk& c = chunks[z * chunksPerDim + x];
    c.heights.assign(h, h + kVerts * kVerts);
    c.mips.clear();
    uint32_t n = kSegs;
    c.mips.emplace_back(n * n);
    for (uint32_t qz = 0; qz < n; ++qz)
        for (uint32_t qx = 0; qx < n; ++qx) {
            uint16_t v[4] = {h[kVerts * qz + qx], h[kVerts * (qz + 1) + qx], h[kVerts * qz + qx + 1], h[kVerts * (qz + 1) + qx + 1]};
            c.mips[0][n * qz + qx] = MinMax{*std::min_element(v, v + 4), *std::max_element(v, v + 4)};
        }
    while (n > 1) {
        const std::vector<MinMax>& src = c.mips.back();
        uint32_t rs = n; n >>= 1;
        std::vector<MinMax> dst(n * n);
        for (uint32_t qz = 0; qz < n; ++qz)
            for (uint32_t qx = 0; qx < n; ++qx) {
                MinMax q[4] = {src[rs * (2 * qz) + 2 * qx], src[rs * (2 * qz + 1) + 2 * qx], src[rs * (2 * qz) + 2 * qx + 1], src[rs * (2 * qz + 1) + 2 * qx + 1]};
                MinMax r = q[0];
                for (int k = 1; k < 4; ++k) { r.mn = std::min(r.mn, q[k].mn); r.mx = std::max(r.mx, q[k].mx); }
                dst[n * qz + qx] = r;
            }
        c.mips.push_back(std::move(dst));
    }
}

// heightmap_collider_component::update — heightmap_collider.cpp:14-19
void Heightmap::update(vec3 corner, float amplitudeScale) {
    minCorner = corner;
    invAmplitudeScale = 1.f / amplitudeScale;
    heightScale = amplitudeScale / 65535.f;
}

static inline uint32_t toU32(float f) { return (uint32_t)(int64_t)f; }   // x86-64 cvttss2si r64 then truncation
static inline float fracf(float v) { return std::fmod(v, 1.f); }        // math.h:40

// heightmap_collider_chunk::getHeightAt — heightmap_collider.cpp:116-153
static float chunkHeightAt(const Heightmap::Chunk& c, float cx, float cz, float heightScale, float heightOffset) {
    if (c.heights.empty()) return -FLT_MAX;
    uint32_t x = toU32(cx), z = toU32(cz);
    float relX = cx - (float)x, relZ = cz - (float)z;
    float a = (float)c.heights[kVerts * z + x] * heightScale, b = (float)c.heights[kVerts * (z + 1) + x] * heightScale;
    float cc = (float)c.heights[kVerts * z + x + 1] * heightScale, d = (float)c.heights[kVerts * (z + 1) + x + 1] * heightScale;
    return lerpf(lerpf(a, cc, relX), lerpf(b, d, relX), relZ) + heightOffset;
}
// heightmap_collider_component::getHeightAt — heightmap_collider.cpp:21-40
float Heightmap::heightAt(float wx, float wz) const {
    float cx = (wx - minCorner.x) * invChunkSize, cz = (wz - minCorner.z) * invChunkSize;
    if (cx < 0.f || cz < 0.f || cx >= (float)chunksPerDim || cz >= (float)chunksPerDim) return -FLT_MAX;
    uint32_t chunkX = toU32(cx), chunkZ = toU32(cz);
    return chunkHeightAt(chunks[chunkZ * chunksPerDim + chunkX], fracf(cx) * (float)kSegs, fracf(cz) * (float)kSegs, heightScale, minCorner.y);
}

// The two iterateTrianglesInVolume levels (heightmap_collider.h:153-205 over chunks, 35-118 inside a chunk): every
// triangle whose quad's min/max-height box and x/z extent touch the volume, in the reference's stack (LIFO) order.
template <typename F>
static void trianglesInVolume(const Heightmap& hm, vec3 vmin, vec3 vmax, const F& func) {
    vmin = vmin - hm.minCorner; vmax = vmax - hm.minCorner;
    vmin.x *= hm.invChunkSize; vmin.z *= hm.invChunkSize; vmax.x *= hm.invChunkSize; vmax.z *= hm.invChunkSize;
    const int32_t cpd = (int32_t)hm.chunksPerDim;
    uint32_t minX = (uint32_t)std::max((int32_t)vmin.x, 0), minZ = (uint32_t)std::max((int32_t)vmin.z, 0);
    uint32_t maxX = (uint32_t)std::min(std::max((int32_t)vmax.x, 0), cpd - 1), maxZ = (uint32_t)std::min(std::max((int32_t)vmax.z, 0), cpd - 1);
    vmin.y *= hm.invAmplitudeScale; vmax.y *= hm.invAmplitudeScale;
    const uint32_t volMinY = (uint16_t)toU32(clamp01(vmin.y) * 65535.f), volMaxY = (uint16_t)toU32(clamp01(vmax.y) * 65535.f);
    for (uint32_t z = minZ; z <= maxZ; ++z)
        for (uint32_t x = minX; x <= maxX; ++x) {
            float relMinX = fmax2(vmin.x - (float)x, 0.f), relMinZ = fmax2(vmin.z - (float)z, 0.f);
            float relMaxX = (vmax.x > (float)(x + 1)) ? 1.f : fracf(vmax.x), relMaxZ = (vmax.z > (float)(z + 1)) ? 1.f : fracf(vmax.z);
            uint32_t volMinX = toU32(relMinX * (float)kVerts), volMinZ = toU32(relMinZ * (float)kVerts);
            uint32_t volMaxX = toU32(relMaxX * (float)kVerts), volMaxZ = toU32(relMaxZ * (float)kVerts);
            vec3 chunkMin = vec3((float)x * hm.chunkSize, 0.f, (float)z * hm.chunkSize) + hm.minCorner;
            const Heightmap::Chunk& c = hm.chunks[z * hm.chunksPerDim + x];
            if (c.heights.empty()) continue;
            struct Node { uint16_t mip, x, z; };
            Node stack[64]; uint32_t top = 0;
            stack[top++] = Node{(uint16_t)(c.mips.size() - 1), 0, 0};
            while (top) {
                Node e = stack[--top];
                uint32_t x0 = (uint32_t)e.x << e.mip, z0 = (uint32_t)e.z << e.mip;
                uint32_t x1 = (((uint32_t)e.x + 1u) << e.mip) - 1u, z1 = (((uint32_t)e.z + 1u) << e.mip) - 1u;
                if (x1 < volMinX || x0 > volMaxX) continue;
                if (z1 < volMinZ || z0 > volMaxZ) continue;
                uint32_t dim = kSegs >> e.mip;
                Heightmap::MinMax mm = c.mips[e.mip][e.z * dim + e.x];
                if (mm.mx < volMinY || mm.mn > volMaxY) continue;
                if (e.mip == 0) {
                    auto vertex = [&](uint32_t vx, uint32_t vz) {
                        float h = (float)c.heights[kVerts * vz + vx] * hm.heightScale;
                        return vec3((float)vx * hm.chunkScale, h, (float)vz * hm.chunkScale) + chunkMin;
                    };
                    vec3 pa = vertex(e.x, e.z), pb = vertex(e.x, e.z + 1u), pc = vertex(e.x + 1u, e.z), pd = vertex(e.x + 1u, e.z + 1u);
                    func(pa, pb, pc);
                    func(pc, pb, pd);
                } else {
                    uint16_t m = (uint16_t)(e.mip - 1);
                    stack[top++] = Node{m, (uint16_t)(2 * e.x), (uint16_t)(2 * e.z)};
                    stack[top++] = Node{m, (uint16_t)(2 * e.x), (uint16_t)(2 * e.z + 1)};
                    stack[top++] = Node{m, (uint16_t)(2 * e.x + 1), (uint16_t)(2 * e.z)};
                    stack[top++] = Node{m, (uint16_t)(2 * e.x + 1), (uint16_t)(2 * e.z + 1)};
                }
            }
        }
}

// closestPoint_PointTriangle — bounding_volumes.cpp:1317-1367 (Voronoi regions of the triangle)
static vec3 closestOnTriangle(vec3 p, vec3 a, vec3 b, vec3 c) {
    vec3 ab = b - a, ac = c - a, ap = p - a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) return a;
    vec3 bp = p - b;
    float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) return b;
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { float v = d1 / (d1 - d3); return a + v * ab; }
    vec3 cp = p - c;
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

struct TriContact { vec3 point, normal; float depth; };

// collideSphereVsTriangle — heightmap_collision.cpp:42-76
static bool sphereVsTriangle(vec3 center, float radius, vec3 a, vec3 b, vec3 c, TriContact& out) {
    vec3 closest = closestOnTriangle(center, a, b, c);
    vec3 n = closest - center;
    float sq = squaredLength(n);
    if (!(sq <= radius * radius)) return false;
    float distance;
    if (sq == 0.f) { n = -cross(b - a, c - a); distance = 0.f; }
    else { distance = std::sqrt(sq); n *= 1.f / distance; }
    out.point = closest; out.normal = n; out.depth = radius - distance;
    return true;
}

// getAABBIncidentEdge — heightmap_collision.cpp:6-40
static void boxIncidentEdge(vec3 r, vec3 normal, vec3& ea, vec3& eb) {
    vec3 p = vabs(normal);
    ea = r;
    if (p.x > p.y) eb = (p.y > p.z) ? vec3(r.x, r.y, -r.z) : vec3(r.x, -r.y, r.z);
    else eb = (p.x > p.z) ? vec3(r.x, r.y, -r.z) : vec3(-r.x, r.y, r.z);
    vec3 s(normal.x < 0.f ? -1.f : 1.f, normal.y < 0.f ? -1.f : 1.f, normal.z < 0.f ? -1.f : 1.f);
    ea = ea * s; eb = eb * s;
}

// collideAABBvsTriangle — heightmap_collision.cpp:78-429: 13-axis SAT (9 edge cross products, 3 box faces in both
// directions, the triangle plane), minimum-penetration axis, one contact.
static bool boxVsTriangle(vec3 center, vec3 radius, vec3 a, vec3 b, vec3 c, TriContact& out) {
    a -= center; b -= center; c -= center;
    const vec3 f[3] = {b - a, c - b, a - c};
    float minPen = FLT_MAX; vec3 minNormal; int category = 0;   // 0..2: edge axis of triangle edge k, 3: box face, 4: triangle plane
    for (int axis = 0; axis < 3; ++axis)       // box axis crossed with the triangle edges: x, then y, then z
        for (int k = 0; k < 3; ++k) {
            const vec3 e = f[k];
            const vec3 q = (k == 0) ? c : b;   // edge 0 projects (a, c), edges 1 and 2 project (a, b)
            float p0, p1, r; vec3 n;
            if (axis == 0) { p0 = (a.z * e.y) - (a.y * e.z); p1 = (q.z * e.y) - (q.y * e.z); r = radius.y * std::fabs(e.z) + radius.z * std::fabs(e.y); n = vec3(0.f, -e.z, e.y); }
            else if (axis == 1) { p0 = (a.x * e.z) - (a.z * e.x); p1 = (q.x * e.z) - (q.z * e.x); r = radius.x * std::fabs(e.z) + radius.z * std::fabs(e.x); n = vec3(e.z, 0.f, -e.x); }
            else { p0 = (a.y * e.x) - (a.x * e.y); p1 = (q.y * e.x) - (q.x * e.y); r = radius.x * std::fabs(e.y) + radius.y * std::fabs(e.x); n = vec3(-e.y, e.x, 0.f); }
            float pen = r - fmax2(-fmax2(p0, p1), fmin2(p0, p1));
            if (pen < 0.f) return false;
            float l = length(n);
            pen *= 1.f / l;
            if (pen < minPen) { minPen = pen; minNormal = n * (1.f / l); category = k; }
        }
    for (int axis = 0; axis < 3; ++axis) {
        vec3 n(0.f);
        float pen = fmax2(a[axis], fmax2(b[axis], c[axis])) + radius[axis];
        if (pen < 0.f) return false;
        if (pen < minPen) { minPen = pen; n[axis] = -1.f; minNormal = n; category = 3; }
        pen = radius[axis] - fmin2(a[axis], fmin2(b[axis], c[axis]));
        if (pen < 0.f) return false;
        if (pen < minPen) { minPen = pen; n = vec3(0.f); n[axis] = 1.f; minNormal = n; category = 3; }
    }
    {
        vec3 triNormal = normalize(cross(f[0], f[1]));
        float pen = dot(radius, vabs(triNormal)) - std::fabs(dot(triNormal, a));
        if (pen < 0.f) return false;
        if (pen < minPen) { minPen = pen; minNormal = triNormal; category = 4; }
    }
    vec3 triCenter = (a + b + c) * (1.f / 3.f);
    if (dot(minNormal, triCenter) < 0.f) minNormal = -minNormal;
    vec3 point;
    if (category < 3) {
        vec3 e0, e1;
        boxIncidentEdge(radius, minNormal, e0, e1);
        vec3 ta = (category == 0) ? a : (category == 1) ? b : c;
        vec3 tb = (category == 0) ? b : (category == 1) ? c : a;
        vec3 pa, pb;
        closestPoint_SegmentSegment(e0, e1, ta, tb, pa, pb);
        point = (pa + pb) * 0.5f;
    } else if (category == 3) {
        float da = dot(minNormal, a), db = dot(minNormal, b), dc = dot(minNormal, c);
        vec3 p = (da < db) ? ((da < dc) ? a : c) : ((db < dc) ? b : c);
        point = p + minNormal * (minPen * 0.5f);
    } else {
        vec3 p(minNormal.x < 0.f ? -radius.x : radius.x, minNormal.y < 0.f ? -radius.y : radius.y, minNormal.z < 0.f ? -radius.z : radius.z);
        point = p - minNormal * (minPen * 0.5f);
    }
    point += center;
    out.point = point; out.normal = minNormal; out.depth = minPen;
    return true;
}

// heightmapCollision — heightmap_collision.cpp:509-618 (for the world's heightmap; physics.cpp:1237-1248)
void heightmapCollision(World& w) {
    w.heightmapCollisions = 0; w.heightmapContacts = 0; w.heightmapManifolds = 0;
    if (!w.heightmap) return;
    const Heightmap& hm = *w.heightmap;
    const uint32_t dummy = (uint32_t)w.bodies.size();
    for (uint32_t i = 0; i < w.wc.size(); ++i) {
        const WorldCollider& col = w.wc[i];
        if (col.objectType != MI_OBJECT_RIGID_BODY) continue;
        if (w.aabbs[i].mx.x < w.aabbs[i].mn.x) continue;   // sharded world: a collider of a body this rank does not simulate this step
        const Shape& s = col.s;
        // (cylinders and hulls: the reference's switch has no case for them and then reads an UNINITIALISED `lowestPoint`, heightmap_collision.cpp:533-573 —
        // undefined behaviour, nothing to reproduce.  What the code evidently means — no triangle routine exists for them, the lowest point of the shape
        // is tested against the surface like for every other type — is what runs here: support(0, -1, 0) of the cylinder / hull, one contact.)
        vec3 vmin = w.aabbs[i].mn, vmax = w.aabbs[i].mx;
        vmax.y += 10.f;
        std::vector<TriContact> found;
        auto keep = [&](bool hit, const TriContact& c) { if (hit && found.size() < 255) found.push_back(c); };
        vec3 lowest;
        SupportShape sup{&s, s.type == T_HULL ? &w.hulls[s.hull] : nullptr};
        switch (s.type) {
            case T_CYLINDER: case T_HULL: break;   // no shape-vs-triangle routine in the reference: the lowest point only
            case T_SPHERE:
                trianglesInVolume(hm, vmin, vmax, [&](vec3 a, vec3 b, vec3 c) { TriContact t; keep(sphereVsTriangle(s.a, s.radius, a, b, c, t), t); });
                break;
            case T_CAPSULE: {   // heightmap_collision.cpp:445-471
                vec3 origin = s.a, dir = normalize(s.b - s.a);
                trianglesInVolume(hm, vmin, vmax, [&](vec3 a, vec3 b, vec3 c) {
                    vec3 triNormal = normalize(cross(b - a, c - a));
                    float d = -dot(triNormal, a);
                    float ndotd = dot(dir, triNormal);
                    float t = -(dot(origin, triNormal) + d) / ndotd;
                    vec3 trace = origin + t * dir;
                    vec3 closest = closestOnTriangle(trace, a, b, c);
                    vec3 reference = closestPoint_PointSegment(closest, s.a, s.b);
                    TriContact tc; keep(sphereVsTriangle(reference, s.radius, a, b, c, tc), tc);
                });
            } break;
            case T_AABB: {
                vec3 center = (s.a + s.b) * 0.5f, radius = (s.b - s.a) * 0.5f;   // bounding_box::getCenter / getRadius
                trianglesInVolume(hm, vmin, vmax, [&](vec3 a, vec3 b, vec3 c) { TriContact t; keep(boxVsTriangle(center, radius, a, b, c, t), t); });
            } break;
            default: {   // T_OBB: the triangle goes into the box frame, the contact comes back (heightmap_collision.cpp:492-507)
                quat inv = conjugate(s.rot);
                trianglesInVolume(hm, vmin, vmax, [&](vec3 a, vec3 b, vec3 c) {
                    TriContact t;
                    if (found.size() < 255 && boxVsTriangle(vec3(0.f), s.b, inv * (a - s.a), inv * (b - s.a), inv * (c - s.a), t)) {
                        t.normal = s.rot * t.normal; t.point = s.rot * t.point + s.a;
                        found.push_back(t);
                    }
                });
            } break;
        }
        lowest = support(sup, vec3(0.f, -1.f, 0.f));
        float h = hm.heightAt(lowest.x, lowest.z);
        if (lowest.y < h && found.size() < 255) found.push_back(TriContact{lowest, vec3(0.f, -1.f, 0.f), h - lowest.y});
        if (found.empty()) continue;
        float friction = clamp01(std::sqrt(col.mat.friction * hm.material.friction));
        float restitution = clamp01(fmax2(col.mat.restitution, hm.material.restitution));
        uint32_t fr = ((uint32_t)(friction * 0xFFFF) << 16) | (uint32_t)(restitution * 0xFFFF);
        for (uint32_t j = 0; j < found.size(); ++j) {
            Contact c; c.point = found[j].point; c.penetrationDepth = found[j].depth; c.normal = found[j].normal; c.friction_restitution = fr;
            if ((j & 3u) == 0u) {   // four consecutive contacts of a collider are one manifold {collider, kHeightmapVirtualBase + j / 4}: solved one after the other, each with its own normal
                w.colliderPairs.push_back(Pair{i, kHeightmapVirtualBase + (j >> 2)});
                w.contactCounts.push_back(std::min<uint32_t>(4u, (uint32_t)found.size() - j));
                ++w.heightmapManifolds;
            }
            w.contacts.push_back(c);
            w.bodyPairs.push_back(Pair{col.objectIndex, dummy});
        }
        ++w.heightmapCollisions; w.heightmapContacts += (uint32_t)found.size();
    }
}

}  // namespace ora
