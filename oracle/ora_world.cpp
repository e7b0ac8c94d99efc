// ora_world.cpp — TEST INFRASTRUCTURE (CPU oracle), see ora_world.h header note.
// Step driver, world-space colliders, SAP broad phase, contact solver, integrator, mass
// properties.  Each function cites the reference lines it restates.
#include "ora_world.h"
#include <algorithm>
#include <unordered_set>
#include <cstring>
#include <limits>
#include <cstdio>

namespace ora {

// ---------------------------------------------------------------- math (non-inline)

mat3 invert(const mat3& m) {  // src/core/math.cpp:276-318
    mat3 inv;
    inv.m00 = m.m11 * m.m22 - m.m21 * m.m12;
    inv.m01 = m.m02 * m.m21 - m.m22 * m.m01;
    inv.m02 = m.m01 * m.m12 - m.m11 * m.m02;
    inv.m10 = m.m12 * m.m20 - m.m22 * m.m10;
    inv.m11 = m.m00 * m.m22 - m.m20 * m.m02;
    inv.m12 = m.m02 * m.m10 - m.m12 * m.m00;
    inv.m20 = m.m10 * m.m21 - m.m20 * m.m11;
    inv.m21 = m.m01 * m.m20 - m.m21 * m.m00;
    inv.m22 = m.m00 * m.m11 - m.m10 * m.m01;
    float det = m.m00 * (m.m11 * m.m22 - m.m21 * m.m12) - m.m01 * (m.m10 * m.m22 - m.m20 * m.m12) + m.m02 * (m.m10 * m.m21 - m.m20 * m.m11);
    if (det == 0.f) return mat3::zero();
    det = 1.f / det;
    return inv * det;
}

quat rotateFromTo(vec3 _from, vec3 _to) {  // src/core/math.cpp:538-575
    vec3 from = normalize(_from), to = normalize(_to);
    float d = dot(from, to);
    if (d >= 1.f) return quat(0.f, 0.f, 0.f, 1.f);
    quat q;
    if (d < (1e-6f - 1.f)) {
        vec3 axis = cross(vec3(1.f, 0.f, 0.f), from);
        if (squaredLength(axis) == 0.f) axis = cross(vec3(0.f, 1.f, 0.f), from);
        axis = normalize(axis);
        // quat(axis, angle): src/core/math.h quat ctor: w = cos(angle/2), v = axis*sin(angle/2)
        float h = kPi * 0.5f;
        float s = det_sinf(h), c = det_cosf(h);
        q = normalize(quat(axis.x * s, axis.y * s, axis.z * s, c));
    } else {
        float s = std::sqrt((1.f + d) * 2.f);
        float invs = 1.f / s;
        vec3 c = cross(from, to);
        q.x = c.x * invs; q.y = c.y * invs; q.z = c.z * invs; q.w = s * 0.5f;
        q = normalize(q);
    }
    return q;
}

void getAxisRotation(quat q, vec3& axis, float& angle) {  // src/core/math.cpp:577-593
    float sqLength = squaredLength(q.v());
    if (sqLength > 0.f) {
        angle = 2.f * det_acosf(q.w);
        float invLength = 1.f / std::sqrt(sqLength);
        axis = q.v() * invLength;
    } else {
        angle = 0.f;
        axis = vec3(1.f, 0.f, 0.f);
    }
}

// det_atan2f / det_acosf / det_sinf / det_cosf: ora_det.cpp

uint32_t hash32(uint32_t m) {  // bijective on 32 bits: unique colouring priorities (joints)
    uint32_t h = m * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return h;
}

// Contact-manifold colouring priority: a bijection on 52 bits of the oriented collider pair (A, B)
// (A, B < 2^26), so priorities are unique and independent of the order manifolds are stored in.
uint64_t pairPriority(uint32_t a, uint32_t b) {
    const uint64_t M52 = (1ull << 52) - 1ull;
    uint64_t x = ((uint64_t)a << 26) | (uint64_t)b;
    x ^= x >> 25; x = (x * 0x9E3779B97F4A7ull) & M52;
    x ^= x >> 27; x = (x * 0xC2B2AE3D27D4Full) & M52;
    x ^= x >> 23;
    return x;
}

// ---------------------------------------------------------------- world

World::World() { joints = jointsCreate(); }
World::~World() { jointsDestroy(joints); delete heightmap; for (Cloth* c : cloths) delete c; }

static float sphereVolume(float r) { float sq = r * r; float sqpi = kPi * sq; return 4.f / 3.f * sqpi * r; }  // bounding_volumes.h:34-40

struct PhysProps { mat3 inertia; vec3 cog; float mass; };

// collider_union::calculatePhysicsProperties — src/physics/physics.cpp:1416-1588
static PhysProps calculatePhysicsProperties(const World& w, const Collider& c) {
    PhysProps r; r.inertia = mat3::zero(); r.mass = 0.f;
    const Shape& s = c.local;
    float density = c.mat.density;
    switch (s.type) {
        case T_SPHERE: {
            r.mass = sphereVolume(s.radius) * density;
            r.cog = s.a;
            r.inertia = mat3::identity() * (2.f / 5.f * r.mass * s.radius * s.radius);
        } break;
        case T_CAPSULE: {
            vec3 axis = s.a - s.b;
            if (axis.y < 0.f) axis *= -1.f;
            float height = length(axis);
            axis *= (1.f / height);
            quat rotation = rotateFromTo(vec3(0.f, 1.f, 0.f), axis);
            mat3 rot = quaternionToMat3(rotation);
            float sqRadius = s.radius * s.radius;
            float sqRadiusPI = kPi * sqRadius;
            // capsule.volume(): bounding_volumes.h:49-57
            float volume = (4.f / 3.f * sqRadiusPI * s.radius) + (sqRadiusPI * length(s.a - s.b));
            r.mass = volume * density;
            r.cog = (s.a + s.b) * 0.5f;
            float cylinderMass = density * sqRadiusPI * height;
            float hemiSphereMass = density * 2.f / 3.f * sqRadiusPI * s.radius;
            float sqCapsuleHeight = height * height;
            mat3 I = mat3::zero();
            I.m11 = sqRadius * cylinderMass * 0.5f;
            I.m00 = I.m22 = I.m11 * 0.5f + cylinderMass * sqCapsuleHeight / 12.f;
            float temp0 = hemiSphereMass * 2.f * sqRadius / 5.f;
            I.m11 += temp0 * 2.f;
            float temp1 = height * 0.5f;
            float temp2 = temp0 + hemiSphereMass * (temp1 * temp1 + 3.f / 8.f * sqCapsuleHeight);
            I.m00 += temp2 * 2.f;
            I.m22 += temp2 * 2.f;
            r.inertia = transpose(rot) * I * rot;
        } break;
        case T_CYLINDER: {
            vec3 axis = s.a - s.b;
            if (axis.y < 0.f) axis *= -1.f;
            float height = length(axis);
            axis *= (1.f / height);
            quat rotation = rotateFromTo(vec3(0.f, 1.f, 0.f), axis);
            mat3 rot = quaternionToMat3(rotation);
            float volume = (kPi * s.radius * s.radius) * length(s.a - s.b);  // bounding_volumes.h:66-72
            r.mass = volume * density;
            r.cog = (s.a + s.b) * 0.5f;
            float sqRadius = s.radius * s.radius;
            float sqHeight = height * height;
            mat3 I = mat3::zero();
            I.m11 = sqRadius * r.mass * 0.5f;
            I.m00 = I.m22 = 1.f / 12.f * r.mass * (3.f * sqRadius + sqHeight);
            r.inertia = transpose(rot) * I * rot;
        } break;
        case T_AABB: {
            vec3 d0 = s.b - s.a;
            r.mass = (d0.x * d0.y * d0.z) * density;
            r.cog = (s.a + s.b) * 0.5f;
            vec3 diameter = ((s.b - s.a) * 0.5f) * 2.f;
            r.inertia.m00 = 1.f / 12.f * r.mass * (diameter.y * diameter.y + diameter.z * diameter.z);
            r.inertia.m11 = 1.f / 12.f * r.mass * (diameter.x * diameter.x + diameter.z * diameter.z);
            r.inertia.m22 = 1.f / 12.f * r.mass * (diameter.x * diameter.x + diameter.y * diameter.y);
        } break;
        case T_OBB: {
            vec3 diameter = s.b * 2.f;
            r.mass = (diameter.x * diameter.y * diameter.z) * density;
            r.cog = s.a;
            mat3 I = mat3::zero();
            I.m00 = 1.f / 12.f * r.mass * (diameter.y * diameter.y + diameter.z * diameter.z);
            I.m11 = 1.f / 12.f * r.mass * (diameter.x * diameter.x + diameter.z * diameter.z);
            I.m22 = 1.f / 12.f * r.mass * (diameter.x * diameter.x + diameter.y * diameter.y);
            mat3 rot = quaternionToMat3(s.rot);
            r.inertia = transpose(rot) * I * rot;
        } break;
        case T_HULL: {
            const HullGeometry& g = w.hulls[s.hull];
            const float s60 = 1.f / 60.f, s120 = 1.f / 120.f;
            mat3 Cc;  // mat3(row-major ctor) symmetric
            Cc.m00 = s60; Cc.m01 = s120; Cc.m02 = s120;
            Cc.m10 = s120; Cc.m11 = s60; Cc.m12 = s120;
            Cc.m20 = s120; Cc.m21 = s120; Cc.m22 = s60;
            float totalMass = 0.f; mat3 totalCov = mat3::zero(); vec3 totalCOG(0.f);
            for (size_t f = 0; f + 2 < g.tris.size(); f += 3) {
                vec3 w1 = s.a + s.rot * g.vertices[g.tris[f]];
                vec3 w2 = s.a + s.rot * g.vertices[g.tris[f + 1]];
                vec3 w3 = s.a + s.rot * g.vertices[g.tris[f + 2]];
                mat3 A;
                A.m00 = w1.x; A.m01 = w2.x; A.m02 = w3.x;
                A.m10 = w1.y; A.m11 = w2.y; A.m12 = w3.y;
                A.m20 = w1.z; A.m21 = w2.z; A.m22 = w3.z;
                float detA = determinant(A);
                mat3 cov = detA * A * Cc * transpose(A);
                float volume = 1.f / 6.f * detA;
                float mass = volume;
                vec3 cog = (w1 + w2 + w3) * 0.25f;
                totalMass += mass;
                totalCov = totalCov + cov;
                totalCOG += cog * mass;
            }
            totalCOG = totalCOG / totalMass;
            mat3 Cprime = totalCov - totalMass * outerProduct(totalCOG, totalCOG);
            r.cog = totalCOG;
            r.mass = totalMass * density;
            r.inertia = mat3::identity() * trace(Cprime) - Cprime;
            r.inertia = r.inertia * density;
        } break;
    }
    return r;
}

// rigid_body_component::recalculateProperties — src/physics/rigid_body.cpp:29-81
void World::recalculateProperties() {
    for (RigidBody& rb : bodies) {
        if (rb.invMass == 0.f) continue;  // kinematic
        const Entity& e = entities[rb.entity];
        size_t n = e.colliders.size();
        if (!n) continue;
        std::vector<PhysProps> props(n);
        for (size_t i = 0; i < n; ++i) props[i] = calculatePhysicsProperties(*this, colliders[e.colliders[i]]);
        mat3 inertia = mat3::zero(); vec3 cog(0.f); float mass = 0.f;
        for (size_t i = 0; i < n; ++i) { mass += props[i].mass; cog += props[i].cog * props[i].mass; }
        rb.invMass = 1.f / mass;
        rb.localCOG = cog = cog * rb.invMass;
        for (size_t i = 0; i < n; ++i) {
            vec3 r = props[i].cog - cog;
            inertia = inertia + (props[i].inertia + (dot(r, r) * mat3::identity() - outerProduct(r, r)) * props[i].mass);
        }
        rb.invInertia = invert(inertia);
    }
    dirtyProps = false;
}

// ---------------------------------------------------------------- K1: world-space colliders

static AABB aabbNegInf() { return AABB{vec3(FLT_MAX, FLT_MAX, FLT_MAX), vec3(-FLT_MAX, -FLT_MAX, -FLT_MAX)}; }
static void grow(AABB& bb, vec3 o) {  // bounding_volumes.cpp:25-33
    bb.mn.x = fmin2(bb.mn.x, o.x); bb.mn.y = fmin2(bb.mn.y, o.y); bb.mn.z = fmin2(bb.mn.z, o.z);
    bb.mx.x = fmax2(bb.mx.x, o.x); bb.mx.y = fmax2(bb.mx.y, o.y); bb.mx.z = fmax2(bb.mx.z, o.z);
}
static AABB transformToAABB(vec3 mn, vec3 mx, quat rotation, vec3 translation) {  // bounding_volumes.cpp:58-70
    AABB r = aabbNegInf();
    grow(r, rotation * mn + translation);
    grow(r, rotation * vec3(mx.x, mn.y, mn.z) + translation);
    grow(r, rotation * vec3(mn.x, mx.y, mn.z) + translation);
    grow(r, rotation * vec3(mx.x, mx.y, mn.z) + translation);
    grow(r, rotation * vec3(mn.x, mn.y, mx.z) + translation);
    grow(r, rotation * vec3(mx.x, mn.y, mx.z) + translation);
    grow(r, rotation * vec3(mn.x, mx.y, mx.z) + translation);
    grow(r, rotation * mx + translation);
    return r;
}

// getWorldSpaceColliders — src/physics/physics.cpp:631-756.  World index k <-> creation index Nc-1-k.
static void getWorldSpaceColliders(World& w) {
    uint32_t nc = (uint32_t)w.colliders.size();
    uint32_t dummy = (uint32_t)w.bodies.size();
    w.wc.resize(nc); w.aabbs.resize(nc);
    for (uint32_t k = 0; k < nc; ++k) {
        const Collider& c = w.colliders[nc - 1 - k];
        const Entity& e = w.entities[c.entity];
        WorldCollider& col = w.wc[k]; AABB& bb = w.aabbs[k];
        vec3 tp; quat tr;
        if (w.shard.enabled && e.rb >= 0 && !w.shard.active[e.rb]) {   // sharded world: a body this rank does not simulate this step (k_world_colliders)
            col.objectIndex = (uint32_t)e.rb; col.objectType = MI_OBJECT_RIGID_BODY; col.mat = c.mat; col.s = Shape(); col.s.type = c.local.type;
            bb.mn = vec3(3.0e38f); bb.mx = vec3(-3.0e38f);
            continue;
        }
        if (e.rb >= 0) { tp = w.bodies[e.rb].p1; tr = w.bodies[e.rb].r1; col.objectIndex = (uint32_t)e.rb; col.objectType = MI_OBJECT_RIGID_BODY; }
        else if (e.kind == MI_ENTITY_FORCE_FIELD) { tp = e.position; tr = e.rotation; col.objectIndex = e.kindIndex; col.objectType = MI_OBJECT_FORCE_FIELD; }
        else if (e.kind == MI_ENTITY_TRIGGER) { tp = e.position; tr = e.rotation; col.objectIndex = e.kindIndex; col.objectType = MI_OBJECT_TRIGGER; }
        else { tp = e.position; tr = e.rotation; col.objectIndex = dummy; col.objectType = MI_OBJECT_STATIC_COLLIDER; }
        col.mat = c.mat;
        col.s = Shape(); col.s.type = c.local.type;
        const Shape& l = c.local;
        switch (l.type) {
            case T_SPHERE: {
                vec3 center = tp + tr * l.a;
                bb.mn = center - vec3(l.radius); bb.mx = center + vec3(l.radius);  // fromCenterRadius
                col.s.a = center; col.s.radius = l.radius;
            } break;
            case T_CAPSULE: {
                vec3 posA = tr * l.a + tp, posB = tr * l.b + tp;
                vec3 r3(l.radius);
                bb = aabbNegInf();
                grow(bb, posA + r3); grow(bb, posA - r3); grow(bb, posB + r3); grow(bb, posB - r3);
                col.s.a = posA; col.s.b = posB; col.s.radius = l.radius;
            } break;
            case T_CYLINDER: {
                vec3 posA = tr * l.a + tp, posB = tr * l.b + tp;
                vec3 a = posB - posA; float aa = dot(a, a);
                float x = 1.f - a.x * a.x / aa, y = 1.f - a.y * a.y / aa, z = 1.f - a.z * a.z / aa;
                x = std::sqrt(fmax2(0.f, x)); y = std::sqrt(fmax2(0.f, y)); z = std::sqrt(fmax2(0.f, z));
                vec3 ev = l.radius * vec3(x, y, z);
                bb.mn = vmin(posA - ev, posB - ev); bb.mx = vmax(posA + ev, posB + ev);
                col.s.a = posA; col.s.b = posB; col.s.radius = l.radius;
            } break;
            case T_AABB: {
                bb = transformToAABB(l.a, l.b, tr, tp);
                if (tr == quat(0.f, 0.f, 0.f, 1.f)) { col.s.a = bb.mn; col.s.b = bb.mx; }
                else {  // promoted to OBB (physics.cpp:725-733; transformToOBB bounding_volumes.cpp:72-79)
                    col.s.type = T_OBB;
                    col.s.a = tr * ((l.a + l.b) * 0.5f) + tp;
                    col.s.b = (l.b - l.a) * 0.5f;
                    col.s.rot = tr;
                }
            } break;
            case T_OBB: {
                // obb.transformToAABB: bounding_volumes.cpp:134-138; transformToOBB: 140-143
                bb = transformToAABB(-l.b, l.b, tr * l.rot, tr * l.a + tp);
                col.s.rot = tr * l.rot; col.s.a = tr * l.a + tp; col.s.b = l.b;
            } break;
            case T_HULL: {
                const HullGeometry& g = w.hulls[l.hull];
                quat rotation = tr * l.rot;
                vec3 position = tr * l.a + tp;
                bb = transformToAABB(g.aabbMin, g.aabbMax, rotation, position);
                col.s.rot = rotation; col.s.a = position; col.s.hull = l.hull;
            } break;
        }
    }
}

// ---------------------------------------------------------------- broad phase

static inline bool aabbVsAABB(const AABB& a, const AABB& b) {  // bounding_volumes.h:352-358
    if (a.mx.x < b.mn.x || a.mn.x > b.mx.x) return false;
    if (a.mx.y < b.mn.y || a.mn.y > b.mx.y) return false;
    if (a.mx.z < b.mn.z || a.mn.z > b.mx.z) return false;
    return true;
}

// broadphase — src/physics/collision_broad.cpp:297-447 with determineOverlapsScalar (87-166).
static void broadphaseReference(World& w) {
    uint32_t nc = (uint32_t)w.colliders.size();
    w.bpPairs.clear();
    if (!nc) return;
    uint32_t axis = w.sortingAxis;
    vec3 s(0.f), s2(0.f);
    for (uint32_t index = 0; index < nc; ++index) {   // view iterates back to front: index <-> creation nc-1-index
        uint32_t creation = nc - 1 - index;
        const AABB& bb = w.aabbs[index];
        uint32_t st = w.startEndpoint[creation], en = w.endEndpoint[creation];
        w.endpoints[st].value = bb.mn[axis]; w.endpoints[en].value = bb.mx[axis];
        w.endpoints[st].colliderIndex = index; w.endpoints[en].colliderIndex = index;
        vec3 center = (bb.mn + bb.mx) * 0.5f;
        s += center; s2 += center * center;
    }
    uint32_t ne = nc * 2;
    std::vector<SapEndpoint>& ep = w.endpoints;
    for (uint32_t i = 1; i < ne; ++i) {  // insertion sort (386-398)
        SapEndpoint key = ep[i];
        uint32_t j = i - 1;
        while (j != UINT32_MAX && ep[j].value > key.value) { ep[j + 1] = ep[j]; j = j - 1; }
        ep[j + 1] = key;
    }
    // sweep (87-166): swap-remove active list, pairs {new, active[k]}
    std::vector<uint32_t> active; active.reserve(nc);
    std::vector<uint32_t> posInActive(nc);
    for (uint32_t i = 0; i < ne; ++i) {
        const SapEndpoint& e = ep[i];
        if (e.start) {
            const AABB& a = w.aabbs[e.colliderIndex];
            for (uint32_t k = 0; k < active.size(); ++k)
                if (aabbVsAABB(a, w.aabbs[active[k]])) w.bpPairs.push_back(Pair{e.colliderIndex, active[k]});
            posInActive[e.colliderIndex] = (uint32_t)active.size();
            active.push_back(e.colliderIndex);
        } else {
            uint32_t pos = posInActive[e.colliderIndex];
            uint32_t last = active.back();
            posInActive[last] = pos;
            active[pos] = last;
            active.pop_back();
        }
    }
    for (uint32_t i = 0; i < ne; ++i) {  // fix up indirections (421-440)
        if (ep[i].start) w.startEndpoint[ep[i].creation] = i; else w.endEndpoint[ep[i].creation] = i;
    }
    vec3 variance = s2 - s * s / (float)nc;
    w.sortingAxis = (variance.x > variance.y) ? ((variance.x > variance.z) ? 0 : 2) : ((variance.y > variance.z) ? 1 : 2);
}

// Centre statistics of the canonical schedule's next sweep axis (same integers on the GPU: k_bp_prepare / k_pair_finish).
// The reference sums centres and squared centres in float, sequentially (collision_broad.cpp:376-384, 443-444); a parallel machine
// needs a statistic whose value does not depend on the order — or on the PARTITION: in a sharded world (include/mi_shard.h) every
// rank sums the colliders it owns and the sums are added over the ranks.  So the centre is quantised to 1/1024 m (clamped to
// +-2^20 m) and q, q^2 are added as INTEGERS: S1 (two's complement in 64 bits), q^2 split into its low 32 bits and the rest
// (S2lo, S2hi; 2^26 colliders cannot overflow either).  Variance order = order of n * S2 - S1^2, compared exactly in 128 bits.
void axisTerms(float c, uint64_t out[3]) {
    const float lim = 1048576.f;
    c = (c > -lim) ? c : -lim;          // (a NaN centre counts as -2^20 on both sides)
    c = (c < lim) ? c : lim;
    const long long q = (long long)rintf(c * 1024.f);
    const uint64_t sq = (uint64_t)(q * q);
    out[0] = (uint64_t)q; out[1] = sq & 0xFFFFFFFFull; out[2] = sq >> 32;
}
uint32_t axisFromSums(const uint64_t s[9], uint32_t n) {
    unsigned __int128 var[3];
    for (int a = 0; a < 3; ++a) {
        const long long s1 = (long long)s[a];
        const unsigned __int128 s2 = ((unsigned __int128)s[6 + a] << 32) + (unsigned __int128)s[3 + a];
        const unsigned __int128 m = (unsigned __int128)(s1 < 0 ? (unsigned long long)(-s1) : (unsigned long long)s1);
        const unsigned __int128 ns2 = (unsigned __int128)n * s2, sq = m * m;
        var[a] = ns2 > sq ? ns2 - sq : 0;     // (>= 0 by Cauchy-Schwarz; a partial, unreduced set of sums may violate it)
    }
    return (var[0] > var[1]) ? ((var[0] > var[2]) ? 0u : 2u) : ((var[1] > var[2]) ? 1u : 2u);   // shape of collision_broad.cpp:443-444
}
// Which colliders a world counts: all of them — or, sharded, those of the bodies it OWNS plus (rank 0 only) the colliders without a
// rigid body (statics, triggers, force fields: replicated on every rank), so that the sum over the ranks counts every collider once.
static void canonicalAxisSums(const World& w, uint64_t s[9]) {
    for (int c = 0; c < 9; ++c) s[c] = 0;
    const uint32_t n = (uint32_t)w.aabbs.size();
    for (uint32_t i = 0; i < n; ++i) {
        if (w.shard.enabled) {
            const WorldCollider& col = w.wc[i];
            const bool counted = col.objectType == MI_OBJECT_RIGID_BODY ? w.shard.active[col.objectIndex] == 1 : w.shard.desc.rank == 0;
            if (!counted) continue;
        }
        for (int c = 0; c < 3; ++c) {
            uint64_t t[3]; axisTerms((w.aabbs[i].mn[c] + w.aabbs[i].mx[c]) * 0.5f, t);
            s[c] += t[0]; s[3 + c] += t[1]; s[6 + c] += t[2];
        }
    }
}

// Canonical pair set: every unordered collider pair whose AABBs overlap (closed intervals), found
// by an independent sort-and-sweep; equals the SAP set except for exact end==start ties.
static void broadphaseCanonical(World& w) {
    uint32_t nc = (uint32_t)w.colliders.size();
    w.bpPairs.clear();
    if (!nc) return;
    uint32_t axis = w.sortingAxis;
    std::vector<uint32_t> order(nc);
    for (uint32_t i = 0; i < nc; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return w.aabbs[a].mn[axis] < w.aabbs[b].mn[axis]; });
    for (uint32_t i = 0; i < nc; ++i) {
        const AABB& a = w.aabbs[order[i]];
        for (uint32_t j = i + 1; j < nc; ++j) {
            const AABB& b = w.aabbs[order[j]];
            if (b.mn[axis] > a.mx[axis]) break;
            if (aabbVsAABB(a, b)) w.bpPairs.push_back(Pair{order[i], order[j]});
        }
    }
    // the next sweep axis.  Sharded: from this rank's own sums until the caller hands in the sums over all ranks (ora_world_shard_set_axis_sums)
    canonicalAxisSums(w, w.axisSums);
    w.sortingAxis = axisFromSums(w.axisSums, nc);
}

// ---------------------------------------------------------------- narrow phase driver

static inline uint32_t bucketOf(int ta, int tb) { return (uint32_t)(ta * 6 - ta * (ta - 1) / 2 + (tb - ta)); }

static uint32_t packMaterial(const Material& a, const Material& b) {  // collision_narrow.cpp:2232-2238
    float friction = clamp01(std::sqrt(a.friction * b.friction));
    float restitution = clamp01(fmax2(a.restitution, b.restitution));
    return ((uint32_t)(friction * 0xFFFF) << 16) | (uint32_t)(restitution * 0xFFFF);
}

static void emitManifold(World& w, const ContactManifold& m, uint32_t a, uint32_t b) {  // writeScalarContact 2221-2253
    const WorldCollider& A = w.wc[a]; const WorldCollider& B = w.wc[b];
    uint32_t fr = packMaterial(A.mat, B.mat);
    w.colliderPairs.push_back(Pair{a, b});
    w.contactCounts.push_back((uint8_t)m.numContacts);
    for (uint32_t i = 0; i < m.numContacts; ++i) {
        Contact c; c.normal = m.normal; c.penetrationDepth = m.depths[i]; c.point = m.points[i]; c.friction_restitution = fr;
        w.contacts.push_back(c);
        w.bodyPairs.push_back(Pair{A.objectIndex, B.objectIndex});
    }
}

// Prune / orient (collision_narrow.cpp:2346-2395).  Returns false if the pair generates no collision.
static bool pruneAndOrient(const World& w, Pair& p, bool& collision) {
    const WorldCollider* A = &w.wc[p.a]; const WorldCollider* B = &w.wc[p.b];
    if (A->objectType != MI_OBJECT_RIGID_BODY && B->objectType != MI_OBJECT_RIGID_BODY) return false;
    if (A->objectType == MI_OBJECT_RIGID_BODY && B->objectType == MI_OBJECT_RIGID_BODY && A->objectIndex == B->objectIndex) return false;
    if (!(A->s.type < B->s.type)) { std::swap(p.a, p.b); std::swap(A, B); }
    collision = (A->objectType == MI_OBJECT_RIGID_BODY && B->objectType == MI_OBJECT_RIGID_BODY)
             || A->objectType == MI_OBJECT_STATIC_COLLIDER || B->objectType == MI_OBJECT_STATIC_COLLIDER;
    return true;
}

// narrowphase — src/physics/collision_narrow.cpp:2328-2603 (collision pairs only; trigger /
// force-field overlap lists are SURVEY §8(f) item 4).
// overlapCheck's bookkeeping (collision_narrow.cpp:1586-1606): which side is the rigid body.
static void pushInteraction(World& w, Pair p) {
    const WorldCollider& A = w.wc[p.a]; const WorldCollider& B = w.wc[p.b];
    if (!overlapCheck(w, A, B)) return;
    Interaction in;
    if (A.objectType == MI_OBJECT_RIGID_BODY) { in.rigidBodyIndex = A.objectIndex; in.otherIndex = B.objectIndex; in.otherType = B.objectType; in.rbCollider = p.a; in.otherCollider = p.b; }
    else { in.rigidBodyIndex = B.objectIndex; in.otherIndex = A.objectIndex; in.otherType = A.objectType; in.rbCollider = p.b; in.otherCollider = p.a; }
    w.interactions.push_back(in);
}

static void narrowphaseReference(World& w) {
    w.colliderPairs.clear(); w.contactCounts.clear(); w.contacts.clear(); w.bodyPairs.clear(); w.interactions.clear();
    std::vector<Pair> buckets[21], ibuckets[21];
    for (Pair p : w.bpPairs) {
        bool collision;
        if (!pruneAndOrient(w, p, collision)) continue;
        (collision ? buckets : ibuckets)[bucketOf(w.wc[p.a].s.type, w.wc[p.b].s.type)].push_back(p);
    }
    for (int bk = 0; bk < 21; ++bk) for (const Pair& p : ibuckets[bk]) pushInteraction(w, p);   // collision_narrow.cpp:2573-2600 (after the collision tests there; independent of them)
    for (int bk = 0; bk < 21; ++bk)
        for (const Pair& p : buckets[bk]) {
            ContactManifold m;
            if (intersect(w, w.wc[p.a], w.wc[p.b], m)) emitManifold(w, m, p.a, p.b);
        }
}

// Canonical order: orientation identical to what the SAP sweep produces ({new, active} then the
// type swap), pairs sorted by (bucket, A, B).
static void narrowphaseCanonical(World& w, uint32_t axisUsed) {
    w.colliderPairs.clear(); w.contactCounts.clear(); w.contacts.clear(); w.bodyPairs.clear(); w.interactions.clear();
    std::vector<Pair> ipairs;
    std::vector<uint64_t> keys; keys.reserve(w.bpPairs.size());
    std::unordered_set<uint64_t> debugListed;
    if (w.debugOrderPending) for (const Pair& p : w.debugOrder) debugListed.insert(((uint64_t)p.a << 32) | p.b);
    for (Pair p : w.bpPairs) {
        // reconstruct {new, active}: new = later start on the sweep axis; tie -> later created = smaller world index
        float ma = w.aabbs[p.a].mn[axisUsed], mb = w.aabbs[p.b].mn[axisUsed];
        bool aIsNew = (ma > mb) || (ma == mb && p.a < p.b);
        Pair q = aIsNew ? Pair{p.a, p.b} : Pair{p.b, p.a};
        bool collision;
        if (!pruneAndOrient(w, q, collision)) continue;
        if (!collision) { ipairs.push_back(q); continue; }
        // ora_debug_set_solve_order: the caller's list also ORIENTS a pair of equal shape type whose AABB starts tie exactly on the sweep axis —
        // there the reference's orientation follows the history of its persistent endpoint array (stable insertion sort), which no rule reproduces
        if (w.debugOrderPending && w.wc[q.a].s.type == w.wc[q.b].s.type && debugListed.count(((uint64_t)q.b << 32) | q.a) && !debugListed.count(((uint64_t)q.a << 32) | q.b)) std::swap(q.a, q.b);
        uint64_t bk = bucketOf(w.wc[q.a].s.type, w.wc[q.b].s.type);
        keys.push_back((bk << 58) | ((uint64_t)q.a << 29) | (uint64_t)q.b);
    }
    std::sort(keys.begin(), keys.end());
    for (uint64_t k : keys) {
        uint32_t a = (uint32_t)((k >> 29) & 0x1FFFFFFFu), b = (uint32_t)(k & 0x1FFFFFFFu);
        ContactManifold m;
        if (intersect(w, w.wc[a], w.wc[b], m)) emitManifold(w, m, a, b);
    }
    for (const Pair& p : ipairs) pushInteraction(w, p);
    // canonical order of the interactions = the order the device applies them in: (rigid body, other collider, body collider)
    std::sort(w.interactions.begin(), w.interactions.end(), [](const Interaction& x, const Interaction& y) {
        if (x.rigidBodyIndex != y.rigidBodyIndex) return x.rigidBodyIndex < y.rigidBodyIndex;
        if (x.otherCollider != y.otherCollider) return x.otherCollider < y.otherCollider;
        return x.rbCollider < y.rbCollider;
    });
}

// ---------------------------------------------------------------- integrator

// applyGravityAndIntegrateForces — src/physics/rigid_body.cpp:95-124
static void applyGravityAndIntegrateForces(RigidBody& rb, GlobalState& g, float dt) {
    g.rotation = rb.r1;
    g.position = rb.p1 + rb.r1 * rb.localCOG;
    mat3 rot = quaternionToMat3(g.rotation);
    g.invInertia = rot * rb.invInertia * transpose(rot);
    g.invMass = rb.invMass;
    if (rb.invMass > 0.f) rb.forceAccumulator.y += (-9.81f / rb.invMass * rb.gravityFactor);
    vec3 linAcc = rb.forceAccumulator * rb.invMass;
    vec3 angAcc = g.invInertia * rb.torqueAccumulator;
    rb.linearVelocity += linAcc * dt;
    rb.angularVelocity += angAcc * dt;
    rb.linearVelocity *= 1.f / (1.f + dt * rb.linearDamping);
    rb.angularVelocity *= 1.f / (1.f + dt * rb.angularDamping);
    g.linearVelocity = rb.linearVelocity;
    g.angularVelocity = rb.angularVelocity;
    g.localCOG = rb.localCOG;
}

// integrateVelocity — src/physics/rigid_body.cpp:126-142
static void integrateVelocity(RigidBody& rb, const GlobalState& g, float dt) {
    rb.linearVelocity = g.linearVelocity;
    rb.angularVelocity = g.angularVelocity;
    quat deltaRot(0.5f * rb.angularVelocity.x, 0.5f * rb.angularVelocity.y, 0.5f * rb.angularVelocity.z, 0.f);
    deltaRot = deltaRot * g.rotation;
    quat rotation = normalize(g.rotation + (deltaRot * dt));
    vec3 position = g.position + rb.linearVelocity * dt;
    rb.forceAccumulator = vec3(0.f); rb.torqueAccumulator = vec3(0.f);
    rb.r1 = rotation;
    rb.p1 = position - rotation * rb.localCOG;
}

// ---------------------------------------------------------------- contact solver

// initializeCollisionVelocityConstraints — src/physics/constraints.cpp:3307-3379
static void initContact(const World& w, uint32_t id, float dt, CollisionConstraint& c) {
    const Contact& contact = w.contacts[id];
    const GlobalState& rbA = w.rb[w.bodyPairs[id].a];
    const GlobalState& rbB = w.rb[w.bodyPairs[id].b];
    float invDt = 1.f / dt;
    c.impulseInNormalDir = 0.f; c.impulseInTangentDir = 0.f;
    c.relGlobalAnchorA = contact.point - rbA.position;
    c.relGlobalAnchorB = contact.point - rbB.position;
    vec3 anchorVelocityA = rbA.linearVelocity + cross(rbA.angularVelocity, c.relGlobalAnchorA);
    vec3 anchorVelocityB = rbB.linearVelocity + cross(rbB.angularVelocity, c.relGlobalAnchorB);
    vec3 relVelocity = anchorVelocityB - anchorVelocityA;
    c.tangent = relVelocity - dot(contact.normal, relVelocity) * contact.normal;
    c.tangent = noz(c.tangent);
    {
        vec3 crAt = cross(c.relGlobalAnchorA, c.tangent);
        vec3 crBt = cross(c.relGlobalAnchorB, c.tangent);
        float invMassT = rbA.invMass + dot(crAt, rbA.invInertia * crAt) + rbB.invMass + dot(crBt, rbB.invInertia * crBt);
        c.effectiveMassInTangentDir = (invMassT != 0.f) ? (1.f / invMassT) : 0.f;
        c.tangentImpulseToAngularVelocityA = rbA.invInertia * crAt;
        c.tangentImpulseToAngularVelocityB = rbB.invInertia * crBt;
    }
    {
        vec3 crAn = cross(c.relGlobalAnchorA, contact.normal);
        vec3 crBn = cross(c.relGlobalAnchorB, contact.normal);
        float invMassN = rbA.invMass + dot(crAn, rbA.invInertia * crAn) + rbB.invMass + dot(crBn, rbB.invInertia * crBn);
        c.effectiveMassInNormalDir = (invMassN != 0.f) ? (1.f / invMassN) : 0.f;
        c.bias = 0.f;
        if (dt > 1e-5f) {
            float vRel = dot(contact.normal, relVelocity);
            const float slop = -0.001f;
            if (-contact.penetrationDepth < slop && vRel < 0.f) {
                float restitution = (float)(contact.friction_restitution & 0xFFFF) / (float)0xFFFF;
                c.bias = -restitution * vRel - 0.1f * (-contact.penetrationDepth - slop) * invDt;
            }
        }
        c.normalImpulseToAngularVelocityA = rbA.invInertia * crAn;
        c.normalImpulseToAngularVelocityB = rbB.invInertia * crBn;
    }
}

// solveCollisionVelocityConstraints body — src/physics/constraints.cpp:3381-3449
static void solveContact(World& w, uint32_t i, CollisionConstraint& c) {
    const Contact& contact = w.contacts[i];
    GlobalState& rbA = w.rb[w.bodyPairs[i].a];
    GlobalState& rbB = w.rb[w.bodyPairs[i].b];
    if (rbA.invMass == 0.f && rbB.invMass == 0.f) return;
    vec3 vA = rbA.linearVelocity, wA = rbA.angularVelocity, vB = rbB.linearVelocity, wB = rbB.angularVelocity;
    {
        vec3 anchorVelocityA = vA + cross(wA, c.relGlobalAnchorA);
        vec3 anchorVelocityB = vB + cross(wB, c.relGlobalAnchorB);
        vec3 relVelocity = anchorVelocityB - anchorVelocityA;
        float vt = dot(relVelocity, c.tangent);
        float lambda = -c.effectiveMassInTangentDir * vt;
        float friction = (float)(contact.friction_restitution >> 16) / (float)0xFFFF;
        float maxFriction = friction * c.impulseInNormalDir;
        float newImpulse = clampf(c.impulseInTangentDir + lambda, -maxFriction, maxFriction);
        lambda = newImpulse - c.impulseInTangentDir;
        c.impulseInTangentDir = newImpulse;
        vec3 P = lambda * c.tangent;
        vA -= rbA.invMass * P;
        wA -= c.tangentImpulseToAngularVelocityA * lambda;
        vB += rbB.invMass * P;
        wB += c.tangentImpulseToAngularVelocityB * lambda;
    }
    {
        vec3 anchorVelocityA = vA + cross(wA, c.relGlobalAnchorA);
        vec3 anchorVelocityB = vB + cross(wB, c.relGlobalAnchorB);
        vec3 relVelocity = anchorVelocityB - anchorVelocityA;
        float vn = dot(relVelocity, contact.normal);
        float lambda = -c.effectiveMassInNormalDir * (vn - c.bias);
        float impulse = fmax2(c.impulseInNormalDir + lambda, 0.f);
        lambda = impulse - c.impulseInNormalDir;
        c.impulseInNormalDir = impulse;
        vec3 P = lambda * contact.normal;
        vA -= rbA.invMass * P;
        wA -= c.normalImpulseToAngularVelocityA * lambda;
        vB += rbB.invMass * P;
        wB += c.normalImpulseToAngularVelocityB * lambda;
    }
    rbA.linearVelocity = vA; rbA.angularVelocity = wA;
    rbB.linearVelocity = vB; rbB.angularVelocity = wB;
}

// Canonical contact schedule (replaces scheduleConstraintsSIMD's role, constraints.cpp:51-184).
// A manifold that already existed in the previous step (same oriented collider pair) KEEPS its colour — two such
// manifolds sharing a dynamic body had different colours then, so they still do — and only the new manifolds are
// coloured: greedy in descending pairPriority(colliderA, colliderB), each taking the lowest colour free on both of its
// dynamic bodies (invMass != 0); bodies with invMass == 0 never conflict (the reference exempts its dummy body the
// same way, constraints.cpp:81-83).  Colour 64 = overflow, solved sequentially last (and re-coloured next step).
// This is exactly what the device computes (hash-table lookup of the previous colours + Jones-Plassmann rounds over
// the rest).  The history is dropped whenever colliders are added (collider world indices shift).
static void colorManifolds(World& w) {
    uint32_t nm = (uint32_t)w.colliderPairs.size();
    w.manifoldColor.assign(nm, 64);
    std::vector<uint64_t> used(w.rb.size(), 0);
    std::vector<uint32_t> firstContact(nm);
    { uint32_t off = 0; for (uint32_t m = 0; m < nm; ++m) { firstContact[m] = off; off += w.contactCounts[m]; } }
    const uint32_t nc = (uint32_t)w.colliders.size();   // keyed by CREATION index (stable when colliders are added later)
    auto keyOf = [&](uint32_t m) {
        uint32_t b = w.colliderPairs[m].b;
        return ((uint64_t)(nc - 1 - w.colliderPairs[m].a) << 26) | (uint64_t)(b >= kHeightmapVirtualBase ? b : nc - 1 - b);
    };
    // exact seam: a SEAM manifold (all of its dynamic bodies are shared between tiles) takes its colour from [0, MI_SEAM_COLORS), any other one from
    // the colours behind them; the two classes never conflict (they are solved one after the other), and a manifold that changed class is re-coloured
    const bool seamMode = w.seamActive();
    const uint64_t seamRange = (1ull << MI_SEAM_COLORS) - 1ull;
    std::vector<uint8_t> isSeam(nm, 0);
    w.seam.manifolds = 0; w.seam.colors = 0;
    if (seamMode) for (uint32_t m = 0; m < nm; ++m) {
        Pair bp = w.bodyPairs[firstContact[m]];
        const bool dynA = w.rb[bp.a].invMass != 0.f, dynB = w.rb[bp.b].invMass != 0.f;
        const uint32_t idA = dynA && bp.a < w.seam.shared.size() ? w.seam.shared[bp.a] : 0u, idB = dynB && bp.b < w.seam.shared.size() ? w.seam.shared[bp.b] : 0u;
        isSeam[m] = (dynA || dynB) && (!dynA || idA) && (!dynB || idB) && (!(dynA && dynB) || idA == idB);   // ... shared across the SAME border
        w.seam.manifolds += isSeam[m];
        // the margin must cover the reach of a contact: a manifold outside the seam class may only touch bodies this rank owns
        if (!isSeam[m] && w.shard.enabled && ((dynA && w.shard.active[bp.a] == 2) || (dynB && w.shard.active[bp.b] == 2))) ++w.seam.violations;
    }
    std::vector<uint32_t> order;
    for (uint32_t m = 0; m < nm; ++m) {
        auto it = w.prevPairColor.find(keyOf(m));
        if (it != w.prevPairColor.end() && it->second < 64 && (!seamMode || (it->second < MI_SEAM_COLORS) == (isSeam[m] != 0))) {
            uint32_t c = it->second;
            Pair bp = w.bodyPairs[firstContact[m]];
            w.manifoldColor[m] = c;
            if (w.rb[bp.a].invMass != 0.f) used[bp.a] |= (1ull << c);
            if (w.rb[bp.b].invMass != 0.f) used[bp.b] |= (1ull << c);
        } else order.push_back(m);
    }
    std::vector<uint64_t> prio(nm);
    for (uint32_t m : order) prio[m] = pairPriority(w.colliderPairs[m].a, w.colliderPairs[m].b);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return prio[a] > prio[b]; });
    for (uint32_t m : order) {
        Pair bp = w.bodyPairs[firstContact[m]];
        bool dynA = w.rb[bp.a].invMass != 0.f, dynB = w.rb[bp.b].invMass != 0.f;
        uint64_t mask = (dynA ? used[bp.a] : 0) | (dynB ? used[bp.b] : 0);
        if (seamMode) mask |= isSeam[m] ? ~seamRange : seamRange;
        if (~mask == 0) { if (seamMode && isSeam[m]) ++w.seam.violations; continue; }  // overflow colour (a seam manifold there would be solved after the interior: not exact)
        uint32_t c = (uint32_t)__builtin_ctzll(~mask);
        w.manifoldColor[m] = c;
        if (dynA) used[bp.a] |= (1ull << c);
        if (dynB) used[bp.b] |= (1ull << c);
    }
    w.prevPairColor.clear();
    for (uint32_t m = 0; m < nm; ++m) w.prevPairColor[keyOf(m)] = w.manifoldColor[m];
    if (seamMode) for (uint32_t m = 0; m < nm; ++m) if (isSeam[m] && w.manifoldColor[m] < MI_SEAM_COLORS) w.seam.colors = std::max(w.seam.colors, w.manifoldColor[m] + 1u);
}

// getForceFieldStates + handleNonCollisionInteractions — src/physics/physics.cpp:759-787, 952-1039.  Localized force fields add
// their (rotated) force to the accumulator of every body they overlap, once per overlapping collider pair; trigger overlaps are
// de-duplicated per (trigger entity, body entity) and diffed against the previous frame.  Returns the global force field.
static vec3 nonCollisionInteractions(World& w) {
    vec3 globalForceField(0.f);
    std::vector<vec3> localForce(w.forceFieldEntities.size());
    for (size_t i = w.forceFieldEntities.size(); i-- > 0;) {   // EnTT view order: back to front
        const Entity& e = w.entities[w.forceFieldEntities[i]];
        vec3 force = e.rotation * e.force;
        if (!e.colliders.empty()) localForce[i] = force; else globalForceField += force;
    }
    std::vector<uint64_t> overlaps;
    for (const Interaction& in : w.interactions) {
        if (in.otherType == MI_OBJECT_FORCE_FIELD) w.bodies[in.rigidBodyIndex].forceAccumulator += localForce[in.otherIndex];
        else if (in.otherType == MI_OBJECT_TRIGGER)
            overlaps.push_back(((uint64_t)w.triggerEntities[in.otherIndex] << 32) | (uint64_t)w.bodies[in.rigidBodyIndex].entity);
    }
    std::sort(overlaps.begin(), overlaps.end());
    overlaps.erase(std::unique(overlaps.begin(), overlaps.end()), overlaps.end());
    if (w.eventsEnabled) {
        auto emit = [&](uint64_t key, uint32_t type) {
            mi_event e{}; e.type = type; e.entity_a = (uint32_t)(key >> 32); e.entity_b = (uint32_t)key; e.collider_a = e.collider_b = 0xFFFFFFFFu;
            w.events.push_back(e);
        };
        size_t p = 0, t = 0;
        while (p < w.prevTriggerOverlaps.size() && t < overlaps.size()) {
            uint64_t pk = w.prevTriggerOverlaps[p], tk = overlaps[t];
            if (pk == tk) { ++p; ++t; }
            else if (pk < tk) { emit(pk, MI_EVENT_TRIGGER_LEAVE); ++p; }
            else { emit(tk, MI_EVENT_TRIGGER_ENTER); ++t; }
        }
        while (p < w.prevTriggerOverlaps.size()) emit(w.prevTriggerOverlaps[p++], MI_EVENT_TRIGGER_LEAVE);
        while (t < overlaps.size()) emit(overlaps[t++], MI_EVENT_TRIGGER_ENTER);
        w.prevTriggerOverlaps = std::move(overlaps);
    } else w.prevTriggerOverlaps.clear();
    return globalForceField;
}

// handleCollisionCallbacks — src/physics/physics.cpp:1041-1178: sorted merge of the previous and the current frame's
// collision lists; begin events carry the mean contact point / normal and the relative point velocity from rbGlobal.
static void collisionEvents(World& w) {
    const uint32_t nc = (uint32_t)w.colliders.size();
    uint32_t nm = (uint32_t)w.colliderPairs.size();
    std::vector<std::pair<uint64_t, uint32_t>> cur;
    std::vector<uint32_t> firstContact(nm);
    { uint32_t off = 0; for (uint32_t m = 0; m < nm; ++m) { firstContact[m] = off; off += w.contactCounts[m]; } }
    for (uint32_t m = 0; m < nm; ++m) {
        if (w.colliderPairs[m].b >= kHeightmapVirtualBase) continue;   // heightmap pairs raise no events (physics.cpp:1050: colliderB < numColliders)
        cur.push_back({((uint64_t)(nc - 1 - w.colliderPairs[m].a) << 26) | (uint64_t)(nc - 1 - w.colliderPairs[m].b), m});
    }
    std::sort(cur.begin(), cur.end());
    auto emit = [&](uint32_t type, uint64_t key, int m) {
        mi_event e{}; e.type = type;
        e.collider_a = (uint32_t)(key >> 26); e.collider_b = (uint32_t)(key & ((1u << 26) - 1u));
        e.entity_a = w.colliders[e.collider_a].entity; e.entity_b = w.colliders[e.collider_b].entity;
        if (m >= 0) {
            uint32_t n = w.contactCounts[m], c0 = firstContact[m];
            float norm = 1.f / (float)n;
            vec3 point(0.f), normal(0.f);
            for (uint32_t i = 0; i < n; ++i) { point += w.contacts[c0 + i].point; normal += w.contacts[c0 + i].normal; }
            point *= norm; normal *= norm;
            const GlobalState& A = w.rb[w.bodyPairs[c0].a]; const GlobalState& B = w.rb[w.bodyPairs[c0].b];
            vec3 velA = A.linearVelocity + cross(A.angularVelocity, point - A.position);
            vec3 velB = B.linearVelocity + cross(B.angularVelocity, point - B.position);
            vec3 rel = velB - velA;
            e.point[0] = point.x; e.point[1] = point.y; e.point[2] = point.z;
            e.normal[0] = normal.x; e.normal[1] = normal.y; e.normal[2] = normal.z;
            e.relative_velocity[0] = rel.x; e.relative_velocity[1] = rel.y; e.relative_velocity[2] = rel.z;
        }
        w.events.push_back(e);
    };
    size_t p = 0, t = 0;
    while (p < w.prevCollisionKeys.size() && t < cur.size()) {
        uint64_t pk = w.prevCollisionKeys[p], tk = cur[t].first;
        if (pk == tk) { ++p; ++t; }
        else if (pk < tk) { emit(MI_EVENT_COLLISION_END, pk, -1); ++p; }
        else { emit(MI_EVENT_COLLISION_BEGIN, tk, (int)cur[t].second); ++t; }
    }
    while (p < w.prevCollisionKeys.size()) emit(MI_EVENT_COLLISION_END, w.prevCollisionKeys[p++], -1);
    while (t < cur.size()) { emit(MI_EVENT_COLLISION_BEGIN, cur[t].first, (int)cur[t].second); ++t; }
    w.prevCollisionKeys.resize(cur.size());
    for (size_t i = 0; i < cur.size(); ++i) w.prevCollisionKeys[i] = cur[i].first;
}

// ---------------------------------------------------------------- step

// physicsStepInternal — src/physics/physics.cpp:1180-1362
void World::stepInternal(const mi_step_settings& settings, float dt) {
    if (dirtyProps) recalculateProperties();
    uint32_t nb = (uint32_t)bodies.size();
    if (nb == 0 && cloths.empty()) return;   // physics.cpp:1184-1189
    if (nb == 0) {                           // cloth only: the rigid-body stages run over nothing
        vec3 wind = nonCollisionInteractions(*this);
        for (Cloth* c : cloths) { c->applyWindForce(wind); c->simulate(clothIterations[0], clothIterations[1], clothIterations[2], dt, orderMode != 0); }
        return;
    }
    uint32_t axisUsed = sortingAxis;
    std::vector<vec3> cogBefore;
    if (shard.enabled) { shardClassify(); cogBefore.resize(nb); for (uint32_t i = 0; i < nb; ++i) cogBefore[i] = bodies[i].p1 + bodies[i].r1 * bodies[i].localCOG; }
    if (seamActive()) seamClassify();
    getWorldSpaceColliders(*this);
    if (orderMode == 0) { broadphaseReference(*this); narrowphaseReference(*this); }
    else { broadphaseCanonical(*this); narrowphaseCanonical(*this, axisUsed); }
    heightmapCollision(*this);   // physics.cpp:1237-1248: after the narrow phase, into the same contact arrays

    // sharded world: a localized force field acts on a ghost for THIS step's solver-side state only — its accumulator is the owner's business (the product keeps the step's
    // forces in a buffer of their own, bForceStep).  Without the restore below a ghost inside a field gathered the field's force step after step (found by
    // tools/gpu_fuzz_sharded.py; no test scene had a localized field in a sharded world).
    std::vector<vec3> accBefore;
    if (shard.enabled) { accBefore.resize(nb); for (uint32_t i = 0; i < nb; ++i) accBefore[i] = bodies[i].forceAccumulator; }
    vec3 globalForceField = nonCollisionInteractions(*this);   // force fields, triggers (physics.cpp:1253-1256)
    rb.resize(nb + 1);
    for (uint32_t i = nb; i-- > 0;) {                           // back to front (1266-1276)
        if (shard.enabled && shard.active[i] != 1) {            // sharded world: only the owner advances a body; a ghost gets its solver-side state from a copy
            if (shard.active[i] == 2) { RigidBody tmp = bodies[i]; tmp.forceAccumulator += globalForceField; applyGravityAndIntegrateForces(tmp, rb[i], dt); }
            continue;
        }
        bodies[i].forceAccumulator += globalForceField;         // physics.cpp:1273
        applyGravityAndIntegrateForces(bodies[i], rb[i], dt);
    }
    if (shard.enabled) for (uint32_t i = 0; i < nb; ++i) if (shard.active[i] != 1) bodies[i].forceAccumulator = accBefore[i];
    std::memset((void*)&rb[nb], 0, sizeof(GlobalState));  // dummy (1279)
    if (eventsEnabled) collisionEvents(*this); else prevCollisionKeys.clear();

    uint32_t ncontacts = (uint32_t)contacts.size();
    if (debugOrderPending) jointsMarkOrderDirty(*this);
    jointsInitialize(*this, dt);
    std::vector<CollisionConstraint> cc(ncontacts);
    for (uint32_t i = 0; i < ncontacts; ++i) initContact(*this, i, dt, cc[i]);

    uint32_t ncolors = 0;
    std::vector<uint32_t> solveOrder;  // contact ids in solve order
    solveOrder.reserve(ncontacts);
    if (orderMode == 0) {
        for (uint32_t i = 0; i < ncontacts; ++i) solveOrder.push_back(i);
    } else if (debugOrderPending) {   // the caller's manifold order (mi_debug_set_solve_order); the colouring still runs: the history stays what the product's is
        colorManifolds(*this);
        uint32_t nm = (uint32_t)colliderPairs.size();
        std::vector<uint32_t> firstContact(nm);
        { uint32_t off = 0; for (uint32_t m = 0; m < nm; ++m) { firstContact[m] = off; off += contactCounts[m]; } }
        std::unordered_map<uint64_t, uint32_t> byPair;
        for (uint32_t m = 0; m < nm; ++m) byPair[((uint64_t)colliderPairs[m].a << 32) | colliderPairs[m].b] = m;
        debugOrderError = nm != debugOrder.size();
        for (const Pair& p : debugOrder) {
            auto it = byPair.find(((uint64_t)p.a << 32) | p.b);
            if (it == byPair.end()) { debugOrderError = true; continue; }
            for (uint32_t k = 0; k < contactCounts[it->second]; ++k) solveOrder.push_back(firstContact[it->second] + k);
        }
        ncolors = 65;
    } else {
        colorManifolds(*this);
        uint32_t nm = (uint32_t)colliderPairs.size();
        std::vector<uint32_t> firstContact(nm);
        { uint32_t off = 0; for (uint32_t m = 0; m < nm; ++m) { firstContact[m] = off; off += contactCounts[m]; } }
        for (uint32_t c = 0; c <= 64; ++c) {
            bool any = false;
            for (uint32_t m = 0; m < nm; ++m) if (manifoldColor[m] == c) {
                any = true;
                for (uint32_t k = 0; k < contactCounts[m]; ++k) solveOrder.push_back(firstContact[m] + k);
            }
            if (any) ncolors = c + 1;
        }
    }
    for (uint32_t it = 0; it < settings.num_rigid_solver_iterations; ++it) {
        jointsSolveIteration(*this);  // distance, ball, fixed, hinge, cone-twist, slider (constraints.cpp:3764-3769)
        for (uint32_t id : solveOrder) solveContact(*this, id, cc[id]);
        // exact seam: the owners' velocities of the shared bodies replace the ghost copies before the next sweep (the caller's exchange)
        if (shard.enabled && seam.exact && seam.fn) { int rc = seam.fn(seam.user, this, it); if (rc != MI_OK && seam.error == MI_OK) seam.error = rc; }
    }
    if (debugOrderPending) { debugOrderPending = false; debugOrder.clear(); jointsMarkOrderDirty(*this); }
    for (uint32_t i = nb; i-- > 0;) { if (shard.enabled && shard.active[i] != 1) continue; integrateVelocity(bodies[i], rb[i], dt); }
    if (shard.enabled) {
        shard.owned[1] = shard.owned[2] = 0;                    // owner rule: a manifold belongs to the owner of its first dynamic body
        uint32_t off = 0;
        for (uint32_t m = 0; m < (uint32_t)colliderPairs.size(); ++m) {
            Pair bp = bodyPairs[off];
            uint32_t first = (bp.a < nb && bodies[bp.a].invMass != 0.f) ? bp.a : bp.b;
            if (first < nb && shard.active[first] == 1) { ++shard.owned[1]; shard.owned[2] += contactCounts[m]; }
            off += contactCounts[m];
        }
        shardPack(cogBefore);
    }
    for (Cloth* c : cloths) {   // physics.cpp:1352-1358
        c->applyWindForce(globalForceField);
        c->simulate(clothIterations[0], clothIterations[1], clothIterations[2], dt, orderMode != 0);
    }

    counts.num_rigid_bodies = nb;
    counts.num_colliders = (uint32_t)colliders.size();
    counts.num_broadphase_overlaps = (uint32_t)bpPairs.size();
    counts.num_collisions = (uint32_t)colliderPairs.size() - heightmapManifolds + heightmapCollisions;   // one collision per collider on the terrain
    counts.num_contacts = ncontacts;
    counts.num_colors = ncolors;
    counts.sorting_axis = axisUsed;
}

static void syncTransformFromPhysics(World& w) {
    for (RigidBody& b : w.bodies) { Entity& e = w.entities[b.entity]; e.position = b.p1; e.rotation = b.r1; }
}

// physicsStep — src/physics/physics.cpp:1364-1413
void World::step(const mi_step_settings& settings, float dt) {
    if (settings.fixed_frame_rate) {
        const float fixedDt = 1.f / (float)settings.frame_rate;
        timer += dt;
        uint32_t iterations = 0;
        if (timer >= fixedDt) {
            for (RigidBody& b : bodies) { b.p0 = b.p1; b.r0 = b.r1; }
            while (timer >= fixedDt && iterations++ < settings.max_physics_iterations_per_frame) {
                stepInternal(settings, fixedDt);
                timer -= fixedDt;
            }
        }
        if (timer >= fixedDt) timer = std::fmod(timer, fixedDt);
        float t = timer / fixedDt;
        for (RigidBody& b : bodies) {  // lerp(trs) src/core/math.h:675-682 (nlerp on the quaternion)
            Entity& e = entities[b.entity];
            e.position = lerp(b.p0, b.p1, t);
            quat q(b.r0.x + t * (b.r1.x - b.r0.x), b.r0.y + t * (b.r1.y - b.r0.y), b.r0.z + t * (b.r1.z - b.r0.z), b.r0.w + t * (b.r1.w - b.r0.w));
            e.rotation = normalize(q);
        }
    } else {
        stepInternal(settings, dt);
        syncTransformFromPhysics(*this);
    }
}

}  // namespace ora

// game_scene::deleteEntity — src/scene/scene.cpp:124-150 with EnTT's pool semantics: every component pool is a packed array, erase =
// swap with the last element and pop.  Colliders are walked newest first (the entity's linked list), each leaves the SAP
// endpoint array (removeColliderFromBroadphase / removeEndpoint, collision_broad.cpp:42-75: the LAST endpoint moves into the
// freed slot) and the collider pool; then the entity's constraints go (deleteAllConstraintsFromEntity), then its rigid body,
// trigger or force-field component.  Entity ids of the ABI stay valid (the slot becomes a tombstone).  The colour history and
// the previous collision / trigger-overlap lists are dropped: they are keyed by pool positions, which just changed
// (the reference keeps its previous-frame list and would compare shifted indices; documented deviation).
namespace ora {
int World::destroyEntity(uint32_t entity) {
    if (entity >= entities.size() || (uint32_t)entities[entity].kind == MI_ENTITY_DESTROYED) return MI_ERR_INVALID_ARGUMENT;
    Entity& e = entities[entity];
    auto removeEndpoint = [&](uint32_t index) {
        SapEndpoint last = endpoints.back();
        endpoints[index] = last;
        if (last.start) startEndpoint[last.creation] = index; else endEndpoint[last.creation] = index;
        endpoints.pop_back();
    };
    const std::vector<uint32_t> mine = e.colliders;      // newest first
    for (size_t k = 0; k < mine.size(); ++k) {
        // ids of colliders that were moved by earlier removals of this loop have been patched in `live` below
        uint32_t id = entities[entity].colliders[0];
        removeEndpoint(startEndpoint[id]);
        removeEndpoint(endEndpoint[id]);
        const uint32_t last = (uint32_t)colliders.size() - 1;
        entities[entity].colliders.erase(entities[entity].colliders.begin());
        if (id != last) {                                   // the last collider of the pool moves into the freed slot
            colliders[id] = colliders[last];
            startEndpoint[id] = startEndpoint[last]; endEndpoint[id] = endEndpoint[last];
            endpoints[startEndpoint[id]].creation = id; endpoints[endEndpoint[id]].creation = id;
            for (uint32_t& c : entities[colliders[id].entity].colliders) if (c == last) c = id;
        }
        colliders.pop_back(); startEndpoint.pop_back(); endEndpoint.pop_back();
    }
    jointsDestroyOfEntity(*this, entity);
    if (e.rb >= 0) {
        const uint32_t p = (uint32_t)e.rb, last = (uint32_t)bodies.size() - 1;
        if (p != last) { bodies[p] = bodies[last]; entities[bodies[p].entity].rb = (int)p; jointsRemapBody(*this, last, p); }
        bodies.pop_back();
    }
    auto dropFrom = [&](std::vector<uint32_t>& pool) {      // trigger_component / force_field_component pools
        const uint32_t p = e.kindIndex, last = (uint32_t)pool.size() - 1;
        if (p != last) { pool[p] = pool[last]; entities[pool[p]].kindIndex = p; }
        pool.pop_back();
    };
    if (e.kind == MI_ENTITY_TRIGGER) dropFrom(triggerEntities);
    if (e.kind == MI_ENTITY_FORCE_FIELD) dropFrom(forceFieldEntities);
    e.kind = MI_ENTITY_DESTROYED; e.rb = -1; e.colliders.clear();
    prevPairColor.clear(); prevCollisionKeys.clear(); prevTriggerOverlaps.clear();
    dirtyProps = true;
    return MI_OK;
}
}  // namespace ora

// ---------------------------------------------------------------- sharded world (include/mi_shard.h; mirrors k_shard_* of the product)
namespace ora {
static uint32_t mortonCode(uint32_t x, uint32_t z) { uint32_t c = 0; for (uint32_t b = 0; b < 16; ++b) c |= ((x >> b) & 1u) << (2 * b) | ((z >> b) & 1u) << (2 * b + 1); return c; }
static std::vector<uint32_t> tilesInRankOrder(uint32_t tx, uint32_t tz) {
    std::vector<uint32_t> t((size_t)tx * tz);
    for (uint32_t i = 0; i < t.size(); ++i) t[i] = i;
    std::sort(t.begin(), t.end(), [&](uint32_t a, uint32_t b) { uint32_t ca = mortonCode(a % tx, a / tx), cb = mortonCode(b % tx, b / tx); return ca != cb ? ca < cb : a < b; });
    return t;
}
// Tiles are cut by BORDERS (tiles - 1 per axis, ascending; tile i of an axis = [border i, border i + 1) with -inf / +inf beyond the rim):
// uniform when sharding is enabled, moved by ora_world_shard_set_borders (load balance).
static const float kInf = std::numeric_limits<float>::infinity();
static float borderLo(const std::vector<float>& b, int tile) { return tile <= 0 ? -kInf : tile > (int)b.size() ? kInf : b[(size_t)tile - 1]; }   // lower border of `tile` (tile may be one past either rim)
static bool shardOwns(const World::Shard& sh, const std::vector<float>& bx, const std::vector<float>& bz, uint32_t t, float x, float z) {
    const int tx = (int)(t % sh.desc.tiles_x), tz = (int)(t / sh.desc.tiles_x);
    return x >= borderLo(bx, tx) && x < borderLo(bx, tx + 1) && z >= borderLo(bz, tz) && z < borderLo(bz, tz + 1);
}
static bool shardInExtended(const World::Shard& sh, const std::vector<float>& bx, const std::vector<float>& bz, uint32_t t, float x, float z) {
    const int tx = (int)(t % sh.desc.tiles_x), tz = (int)(t / sh.desc.tiles_x);
    const float m = sh.desc.ghost_margin;
    return x >= borderLo(bx, tx) - m && x < borderLo(bx, tx + 1) + m && z >= borderLo(bz, tz) - m && z < borderLo(bz, tz + 1) + m;
}
void World::shardClassify() {
    const uint32_t nb = (uint32_t)bodies.size();
    shard.active.assign(nb, 0); shard.owned[0] = 0;
    jointsIslandRoots(*this, shard.root);                        // an articulated island is owned / ghosted / ignored as ONE: by its root body's centre
    for (uint32_t i = 0; i < nb; ++i) {
        const RigidBody& rbody = bodies[shard.root[i]];
        if (!rbody.shardKnown) continue;                         // a copy that is not current says nothing about where the body is
        vec3 c = rbody.p1 + rbody.r1 * rbody.localCOG;
        bool owned = shardOwns(shard, shard.bordersX, shard.bordersZ, shard.myTile, c.x, c.z);
        shard.active[i] = owned ? 1 : shardInExtended(shard, shard.bordersX, shard.bordersZ, shard.myTile, c.x, c.z) ? 2 : 0;
        shard.owned[0] += owned ? 1u : 0u;
    }
}
void World::shardPack(const std::vector<vec3>& oldCog) {
    const uint32_t nb = (uint32_t)bodies.size();
    for (size_t k = 0; k < shard.peers.size(); ++k) {
        std::vector<float>& msg = shard.sendBuf[k];
        msg.assign((size_t)(shard.capacity + 1u) * MI_SHARD_RECORD_FLOATS, 0.f);
        uint32_t n = 0;
        for (uint32_t i = 0; i < nb; ++i) {
            if (shard.active[i] != 1) continue;
            const RigidBody& b = bodies[i];
            const RigidBody& rbody = bodies[shard.root[i]];
            vec3 cn = rbody.p1 + rbody.r1 * rbody.localCOG, co = oldCog[shard.root[i]];
            bool want = shardInExtended(shard, shard.bordersX, shard.bordersZ, shard.peers[k], cn.x, cn.z) || shardInExtended(shard, shard.bordersX, shard.bordersZ, shard.peers[k], co.x, co.z);
            // borders about to move: also what the neighbour simulates under the NEW borders (it classifies with them from the next step on)
            if (shard.bordersPending) want = want || shardInExtended(shard, shard.nextX, shard.nextZ, shard.peers[k], cn.x, cn.z);
            if (!want) continue;
            if (n < shard.capacity) {
                float* o = msg.data() + (size_t)(n + 1u) * MI_SHARD_RECORD_FLOATS;
                std::memcpy(o, &i, 4);
                o[1] = b.p1.x; o[2] = b.p1.y; o[3] = b.p1.z; o[4] = b.r1.x; o[5] = b.r1.y; o[6] = b.r1.z; o[7] = b.r1.w;
                o[8] = b.linearVelocity.x; o[9] = b.linearVelocity.y; o[10] = b.linearVelocity.z;
                o[11] = b.angularVelocity.x; o[12] = b.angularVelocity.y; o[13] = b.angularVelocity.z;
            }
            ++n;
        }
        std::memcpy(msg.data(), &n, 4);
    }
    // what this rank knows from here on: the bodies it owned; the records about to arrive add the neighbours' (ora_world_shard_import)
    for (uint32_t i = 0; i < nb; ++i) bodies[i].shardKnown = shard.active[i] == 1 ? 1 : 0;
    if (shard.bordersPending) { shard.bordersX = shard.nextX; shard.bordersZ = shard.nextZ; shard.bordersPending = false; }
}
// Which tile border is (x, z) within the margin of, i.e. which neighbour of the tile that contains it sees it as well?  0 = none, else 1 + the border's
// index along x | (1 + index along z) << 16 (the same comparisons as shardInExtended; tiles are at least two margins wide, so at most one per axis).
static uint32_t seamBorderOf(const std::vector<float>& bx, const std::vector<float>& bz, float m, float x, float z) {
    auto near = [&](const std::vector<float>& b, float v) -> uint32_t {
        int t = 0; while (t < (int)b.size() && v >= b[(size_t)t]) ++t;       // tile index along this axis: borders b[t - 1] <= v < b[t]
        if (t > 0 && v < b[(size_t)t - 1] + m) return (uint32_t)t;           // border t - 1
        if (t < (int)b.size() && v >= b[(size_t)t] - m) return (uint32_t)t + 1u;
        return 0u;
    };
    return near(bx, x) | (near(bz, z) << 16);
}
void World::seamClassify() {
    const uint32_t nb = (uint32_t)bodies.size();
    seam.shared.assign(nb, 0); seam.cogStart.resize(nb);
    std::vector<uint32_t> root;
    if (shard.enabled) root = shard.root; else jointsIslandRoots(*this, root);
    const std::vector<float>& bx = shard.enabled ? shard.bordersX : seam.bx;
    const std::vector<float>& bz = shard.enabled ? shard.bordersZ : seam.bz;
    const float m = shard.enabled ? shard.desc.ghost_margin : seam.margin;
    for (uint32_t i = 0; i < nb; ++i) {
        const RigidBody& rbody = bodies[root[i]];
        const vec3 c = rbody.p1 + rbody.r1 * rbody.localCOG;
        seam.cogStart[i] = c;
        seam.shared[i] = (!shard.enabled || shard.active[i] != 0) ? seamBorderOf(bx, bz, m, c.x, c.z) : 0u;   // (a ghost is within the margin of one of this tile's borders)
    }
}
// Borders that even out the body counts: hist = bodies per bin of [lo, hi) along one axis, summed over all ranks.  Border i goes where the
// cumulative count reaches i / tiles of the total (linear inside a bin), then is clamped to what one change may do (see shardBordersValid).
static bool shardBordersValid(const std::vector<float>& cur, const float* nb, uint32_t n, float m) {
    for (uint32_t i = 0; i < n; ++i) {
        if (!(nb[i] == nb[i])) return false;
        if (i > 0 && !(nb[i] - nb[i - 1] > m)) return false;                       // a tile narrower than the margin would need more than its 8 neighbours
        if (i > 0 && nb[i] < cur[i - 1] + m) return false;                         // new owner / ghost holder of a body = old owner's tile or one next to it
        if (i + 1 < n && nb[i] > cur[i + 1] - m) return false;
    }
    return true;
}
}  // namespace ora

// ================================================================= C ABI (mirrors include/mi_physics.h with an ora_ prefix)
using namespace ora;

static void shapeFromDesc(const mi_collider_desc& d, Shape& s) {
    s.type = (int)d.type;
    const float* f = d.shape;
    switch (d.type) {
        case T_SPHERE: s.a = vec3(f[0], f[1], f[2]); s.radius = f[3]; break;
        case T_CAPSULE: case T_CYLINDER: s.a = vec3(f[0], f[1], f[2]); s.b = vec3(f[3], f[4], f[5]); s.radius = f[6]; break;
        case T_AABB: s.a = vec3(f[0], f[1], f[2]); s.b = vec3(f[3], f[4], f[5]); break;
        case T_OBB: s.rot = quat(f[0], f[1], f[2], f[3]); s.a = vec3(f[4], f[5], f[6]); s.b = vec3(f[7], f[8], f[9]); break;
        case T_HULL: s.rot = quat(f[0], f[1], f[2], f[3]); s.a = vec3(f[4], f[5], f[6]); s.hull = d.hull_geometry; break;
    }
}

extern "C" {

MI_API int ora_world_create(int order_mode, World** out) { *out = new World(); (*out)->orderMode = order_mode; return MI_OK; }
MI_API void ora_world_destroy(World* w) { delete w; }

MI_API int ora_entities_create(World* w, uint32_t count, const mi_entity_desc* descs, uint32_t* out_first) {
    if (out_first) *out_first = (uint32_t)w->entities.size();
    for (uint32_t i = 0; i < count; ++i) {
        const mi_entity_desc& d = descs[i];
        Entity e; e.position = vec3(d.position[0], d.position[1], d.position[2]);
        e.rotation = quat(d.rotation[0], d.rotation[1], d.rotation[2], d.rotation[3]); e.kind = (int)d.kind;
        if (d.kind == MI_ENTITY_FORCE_FIELD) { e.kindIndex = (uint32_t)w->forceFieldEntities.size(); w->forceFieldEntities.push_back((uint32_t)w->entities.size()); }
        if (d.kind == MI_ENTITY_TRIGGER) { e.kindIndex = (uint32_t)w->triggerEntities.size(); w->triggerEntities.push_back((uint32_t)w->entities.size()); }
        if (d.kind == MI_ENTITY_DYNAMIC || d.kind == MI_ENTITY_KINEMATIC) {
            RigidBody rb;
            rb.entity = (uint32_t)w->entities.size();
            bool kinematic = d.kind == MI_ENTITY_KINEMATIC;  // rigid_body.cpp:6-27
            rb.invMass = kinematic ? 0.f : 1.f;
            rb.invInertia = kinematic ? mat3::zero() : mat3::identity();
            rb.gravityFactor = d.gravity_factor; rb.linearDamping = d.linear_damping; rb.angularDamping = d.angular_damping;
            rb.localCOG = vec3(0.f);
            rb.linearVelocity = vec3(d.linear_velocity[0], d.linear_velocity[1], d.linear_velocity[2]);
            rb.angularVelocity = vec3(d.angular_velocity[0], d.angular_velocity[1], d.angular_velocity[2]);
            rb.forceAccumulator = vec3(0.f); rb.torqueAccumulator = vec3(0.f);
            rb.p0 = rb.p1 = e.position; rb.r0 = rb.r1 = e.rotation;
            e.rb = (int)w->bodies.size();
            w->bodies.push_back(rb);
        }
        w->entities.push_back(e);
    }
    w->dirtyProps = true;
    return MI_OK;
}
MI_API int ora_entity_create(World* w, const mi_entity_desc* d, uint32_t* out) { return ora_entities_create(w, 1, d, out); }
MI_API int ora_entity_destroy(World* w, uint32_t entity) { return w ? w->destroyEntity(entity) : MI_ERR_INVALID_ARGUMENT; }

MI_API int ora_colliders_add(World* w, uint32_t count, const uint32_t* entities, const mi_collider_desc* descs) {
    for (uint32_t i = 0; i < count; ++i) {
        if (entities[i] >= w->entities.size()) return MI_ERR_INVALID_ARGUMENT;
        Collider c; shapeFromDesc(descs[i], c.local);
        c.mat = Material{descs[i].restitution, descs[i].friction, descs[i].density};
        c.entity = entities[i];
        uint32_t id = (uint32_t)w->colliders.size();
        w->colliders.push_back(c);
        Entity& e = w->entities[entities[i]];
        e.colliders.insert(e.colliders.begin(), id);
        // addColliderToBroadphase — collision_broad.cpp:27-40
        w->startEndpoint.push_back((uint32_t)w->endpoints.size());
        w->endpoints.push_back(SapEndpoint{0.f, id, true, 0});
        w->endEndpoint.push_back((uint32_t)w->endpoints.size());
        w->endpoints.push_back(SapEndpoint{0.f, id, false, 0});
    }
    w->dirtyProps = true;
    return MI_OK;
}
MI_API int ora_collider_add(World* w, uint32_t entity, const mi_collider_desc* d, uint32_t* out) {
    if (out) *out = (uint32_t)w->colliders.size();
    return ora_colliders_add(w, 1, &entity, d);
}

MI_API int ora_hull_geometry_create(World* w, const float* v, uint32_t nv, const uint32_t* t, uint32_t nt, uint32_t* out) {
    HullGeometry g;
    g.aabbMin = vec3(FLT_MAX); g.aabbMax = vec3(-FLT_MAX);
    for (uint32_t i = 0; i < nv; ++i) {
        vec3 p(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
        g.vertices.push_back(p);
        g.aabbMin = vmin(g.aabbMin, p); g.aabbMax = vmax(g.aabbMax, p);
    }
    g.tris.assign(t, t + 3 * nt);
    *out = (uint32_t)w->hulls.size();
    w->hulls.push_back(g);
    return MI_OK;
}

MI_API int ora_constraint_create(World* w, uint32_t type, uint32_t ea, uint32_t eb, const void* pod, uint32_t bytes, uint32_t* out) {
    return jointsAdd(*w, type, ea, eb, pod, bytes, out);
}
MI_API int ora_constraint_update(World* w, uint32_t type, uint32_t id, const void* pod, uint32_t bytes) { return jointsUpdate(*w, type, id, pod, bytes); }
MI_API int ora_constraint_get(World* w, uint32_t type, uint32_t id, void* pod, uint32_t bytes) { return jointsGet(*w, type, id, pod, bytes); }
MI_API int ora_constraint_destroy(World* w, uint32_t type, uint32_t id) { return jointsDestroy(*w, type, id); }
MI_API int ora_constraints_destroy_all(World* w) { jointsDestroyAll(*w); return MI_OK; }
MI_API int ora_entity_destroy_constraints(World* w, uint32_t entity) { return entity < w->entities.size() ? jointsDestroyOfEntity(*w, entity) : MI_ERR_INVALID_ARGUMENT; }
MI_API int ora_constraint_create_from_global(World* w, uint32_t type, uint32_t ea, uint32_t eb, const float* anchor, const float* axis,
                                             float l0, float l1, uint32_t* out) {
    return jointsAddFromGlobal(*w, type, ea, eb, anchor, axis, l0, l1, out);
}

MI_API int ora_entity_apply_force(World* w, uint32_t entity, const float* f, const float* t) {
    if (entity >= w->entities.size() || w->entities[entity].rb < 0) return MI_ERR_INVALID_ARGUMENT;
    RigidBody& rb = w->bodies[w->entities[entity].rb];
    if (f) rb.forceAccumulator += vec3(f[0], f[1], f[2]);
    if (t) rb.torqueAccumulator += vec3(t[0], t[1], t[2]);
    return MI_OK;
}

MI_API int ora_entities_apply_forces(World* w, uint32_t count, const uint32_t* ents, const float* f, const float* t) {
    for (uint32_t i = 0; i < count; ++i) {
        int rc = ora_entity_apply_force(w, ents[i], f ? f + 3 * i : nullptr, t ? t + 3 * i : nullptr);
        if (rc != MI_OK) return rc;
    }
    return MI_OK;
}
MI_API int ora_constraints_update(World* w, uint32_t type, uint32_t count, const uint32_t* ids, const void* pods, uint32_t podBytes) {
    for (uint32_t i = 0; i < count; ++i) {
        int rc = jointsUpdate(*w, type, ids[i], (const char*)pods + (size_t)i * podBytes, podBytes);
        if (rc != MI_OK) return rc;
    }
    return MI_OK;
}
// testPhysicsInteraction — src/physics/physics.cpp:555-629: the closest rigid-body collider along the ray (strict `<`, colliders in
// EnTT view order = newest first) gets force = direction * strength at the hit point.  Ray i only sees the colliders of the
// entities [ranges[2i], ranges[2i+1]) (null: the whole scene).
MI_API int ora_world_test_interactions(World* w, uint32_t count, const float* origins, const float* directions, const float* strengths, const uint32_t* ranges) {
    if (!w || (count && (!origins || !directions))) return MI_ERR_INVALID_ARGUMENT;
    if (w->dirtyProps) w->recalculateProperties();
    for (uint32_t r = 0; r < count; ++r) {
        vec3 ro(origins[3 * r], origins[3 * r + 1], origins[3 * r + 2]), rd(directions[3 * r], directions[3 * r + 1], directions[3 * r + 2]);
        float strength = strengths ? strengths[r] : 1000.f;
        uint32_t lo = ranges ? ranges[2 * r] : 0u, hi = ranges ? ranges[2 * r + 1] : 0xFFFFFFFFu;
        float minT = FLT_MAX; RigidBody* minRB = nullptr; vec3 force, torque;
        for (size_t k = w->colliders.size(); k-- > 0;) {
            const Collider& c = w->colliders[k];
            const Entity& e = w->entities[c.entity];
            if (e.rb < 0 || c.entity < lo || c.entity >= hi) continue;
            RigidBody& rb = w->bodies[e.rb];
            vec3 lo_ = conjugate(rb.r1) * (ro - rb.p1), ld = conjugate(rb.r1) * rd;   // physics_transform1
            float t;
            if (rayVsCollider(*w, c.local, lo_, ld, t) && t < minT) {
                minT = t; minRB = &rb;
                vec3 localHit = lo_ + t * ld;
                vec3 globalHit = rb.r1 * localHit + rb.p1;
                vec3 cog = rb.p1 + rb.r1 * rb.localCOG;
                force = rd * strength;
                torque = cross(globalHit - cog, force);
            }
        }
        if (minRB) { minRB->torqueAccumulator += torque; minRB->forceAccumulator += force; }
    }
    return MI_OK;
}
MI_API int ora_world_step(World* w, const mi_step_settings* s, float dt) {
    w->step(*s, dt);
    if (w->debugOrderError) { w->debugOrderError = false; return MI_ERR_INVALID_ARGUMENT; }   // ora_debug_set_solve_order: the list did not match the step's manifolds
    return MI_OK;
}
MI_API int ora_world_step_fixed(World* w, const mi_step_settings* s, float dt, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) w->stepInternal(*s, dt);
    for (RigidBody& b : w->bodies) { Entity& e = w->entities[b.entity]; e.position = b.p1; e.rotation = b.r1; }
    if (w->debugOrderError) { w->debugOrderError = false; return MI_ERR_INVALID_ARGUMENT; }   // ora_debug_set_solve_order: the list did not match the step's manifolds
    if (w->seam.error != MI_OK) { const int rc = w->seam.error; w->seam.error = MI_OK; return rc; }   // the exact seam's sweep exchange failed
    return MI_OK;
}

MI_API int ora_world_num_entities(World* w, uint32_t* out) { *out = (uint32_t)w->entities.size(); return MI_OK; }
static int getTransforms(World* w, float* p, float* r, uint32_t cap, bool physics) {
    uint32_t n = (uint32_t)w->entities.size();
    if (cap < n) return MI_ERR_CAPACITY;
    for (uint32_t i = 0; i < n; ++i) {
        const Entity& e = w->entities[i];
        vec3 pos = e.position; quat rot = e.rotation;
        if (physics && e.rb >= 0) { pos = w->bodies[e.rb].p1; rot = w->bodies[e.rb].r1; }
        if ((uint32_t)e.kind == MI_ENTITY_DESTROYED) { pos = vec3(0.f); rot = quat(0.f, 0.f, 0.f, 1.f); }
        if (p) { p[3 * i] = pos.x; p[3 * i + 1] = pos.y; p[3 * i + 2] = pos.z; }
        if (r) { r[4 * i] = rot.x; r[4 * i + 1] = rot.y; r[4 * i + 2] = rot.z; r[4 * i + 3] = rot.w; }
    }
    return MI_OK;
}
MI_API int ora_world_get_transforms(World* w, float* p, float* r, uint32_t cap) { return getTransforms(w, p, r, cap, false); }
MI_API int ora_world_get_physics_transforms(World* w, float* p, float* r, uint32_t cap) { return getTransforms(w, p, r, cap, true); }
MI_API int ora_world_get_velocities(World* w, float* lin, float* ang, uint32_t cap) {
    uint32_t n = (uint32_t)w->entities.size();
    if (cap < n) return MI_ERR_CAPACITY;
    for (uint32_t i = 0; i < n; ++i) {
        vec3 v(0.f), a(0.f);
        if (w->entities[i].rb >= 0) { v = w->bodies[w->entities[i].rb].linearVelocity; a = w->bodies[w->entities[i].rb].angularVelocity; }
        if (lin) { lin[3 * i] = v.x; lin[3 * i + 1] = v.y; lin[3 * i + 2] = v.z; }
        if (ang) { ang[3 * i] = a.x; ang[3 * i + 1] = a.y; ang[3 * i + 2] = a.z; }
    }
    return MI_OK;
}
MI_API int ora_world_get_mass_properties(World* w, float* invMass, float* invInertia, float* cog, uint32_t cap) {
    if (w->dirtyProps) w->recalculateProperties();
    uint32_t n = (uint32_t)w->entities.size();
    if (cap < n) return MI_ERR_CAPACITY;
    for (uint32_t i = 0; i < n; ++i) {
        float im = 0.f; mat3 ii = mat3::zero(); vec3 c(0.f);
        if (w->entities[i].rb >= 0) { const RigidBody& b = w->bodies[w->entities[i].rb]; im = b.invMass; ii = b.invInertia; c = b.localCOG; }
        if (invMass) invMass[i] = im;
        if (invInertia) std::memcpy(invInertia + 9 * i, ii.data(), 36);
        if (cog) { cog[3 * i] = c.x; cog[3 * i + 1] = c.y; cog[3 * i + 2] = c.z; }
    }
    return MI_OK;
}
MI_API int ora_world_get_counts(World* w, mi_step_counts* out) { *out = w->counts; return MI_OK; }
MI_API int ora_world_get_contacts(World* w, mi_contact* out, uint32_t cap, uint32_t* count) {
    uint32_t n = (uint32_t)w->contacts.size();
    *count = n;
    if (!out) return MI_OK;
    if (cap < n) return MI_ERR_CAPACITY;
    uint32_t ci = 0;
    for (uint32_t m = 0; m < w->colliderPairs.size(); ++m)
        for (uint32_t k = 0; k < w->contactCounts[m]; ++k, ++ci) {
            const Contact& c = w->contacts[ci]; mi_contact& o = out[ci];
            o.point[0] = c.point.x; o.point[1] = c.point.y; o.point[2] = c.point.z; o.penetration_depth = c.penetrationDepth;
            o.normal[0] = c.normal.x; o.normal[1] = c.normal.y; o.normal[2] = c.normal.z; o.friction_restitution = c.friction_restitution;
            o.collider_a = w->colliderPairs[m].a; o.collider_b = w->colliderPairs[m].b >= kHeightmapVirtualBase ? 0xFFFFFFFFu : w->colliderPairs[m].b;
            o.body_a = w->bodyPairs[ci].a; o.body_b = w->bodyPairs[ci].b;
        }
    return MI_OK;
}
MI_API int ora_cloth_create(World* w, const mi_cloth_desc* d, uint32_t* out) {
    if (!w || !d || d->grid_size_x < 2 || d->grid_size_y < 2 || !(d->total_mass > 0.f) || !(d->stiffness > 0.f)) return MI_ERR_INVALID_ARGUMENT;
    if (out) *out = (uint32_t)w->cloths.size();
    w->cloths.push_back(new Cloth(*d));
    return MI_OK;
}
MI_API int ora_cloth_set_fixed_vertices(World* w, uint32_t cloth, const float* p, const float* r, uint32_t moveRigid) {
    if (!w || cloth >= w->cloths.size() || !p || !r) return MI_ERR_INVALID_ARGUMENT;
    w->cloths[cloth]->setFixedVertices(vec3(p[0], p[1], p[2]), quat(r[0], r[1], r[2], r[3]), moveRigid != 0);
    return MI_OK;
}
MI_API int ora_cloth_set_properties(World* w, uint32_t cloth, float totalMass, float stiffness, float damping, float gravityFactor) {
    if (!w || cloth >= w->cloths.size()) return MI_ERR_INVALID_ARGUMENT;
    Cloth& c = *w->cloths[cloth];
    c.totalMass = totalMass; c.stiffness = stiffness; c.damping = damping; c.gravityFactor = gravityFactor;
    return MI_OK;
}
MI_API int ora_cloth_get_state(World* w, uint32_t cloth, float* pos, float* vel, uint32_t cap) {
    if (!w || cloth >= w->cloths.size()) return MI_ERR_INVALID_ARGUMENT;
    const Cloth& c = *w->cloths[cloth];
    if (cap < c.positions.size()) return MI_ERR_CAPACITY;
    for (size_t i = 0; i < c.positions.size(); ++i) {
        if (pos) { pos[3 * i] = c.positions[i].x; pos[3 * i + 1] = c.positions[i].y; pos[3 * i + 2] = c.positions[i].z; }
        if (vel) { vel[3 * i] = c.velocities[i].x; vel[3 * i + 1] = c.velocities[i].y; vel[3 * i + 2] = c.velocities[i].z; }
    }
    return MI_OK;
}
MI_API int ora_world_set_cloth_iterations(World* w, uint32_t v, uint32_t p, uint32_t d) {
    if (!w) return MI_ERR_INVALID_ARGUMENT;
    w->clothIterations[0] = v; w->clothIterations[1] = p; w->clothIterations[2] = d;
    return MI_OK;
}
MI_API int ora_heightmap_create(World* w, uint32_t chunksPerDim, float chunkSize, float restitution, float friction) {
    if (!w || !chunksPerDim || !(chunkSize > 0.f) || w->heightmap) return MI_ERR_INVALID_ARGUMENT;
    w->heightmap = new Heightmap(chunksPerDim, chunkSize, Material{restitution, friction, 0.f});
    return MI_OK;
}
MI_API int ora_heightmap_set_chunk_heights(World* w, uint32_t x, uint32_t z, const uint16_t* heights) {
    if (!w || !w->heightmap || !heights || x >= w->heightmap->chunksPerDim || z >= w->heightmap->chunksPerDim) return MI_ERR_INVALID_ARGUMENT;
    w->heightmap->setHeights(x, z, heights);
    return MI_OK;
}
MI_API int ora_heightmap_update(World* w, const float* minCorner, float amplitudeScale) {
    if (!w || !w->heightmap || !minCorner) return MI_ERR_INVALID_ARGUMENT;
    w->heightmap->update(vec3(minCorner[0], minCorner[1], minCorner[2]), amplitudeScale);
    return MI_OK;
}
MI_API int ora_heightmap_get_height(World* w, float x, float z, float* out) {
    if (!w || !w->heightmap || !out) return MI_ERR_INVALID_ARGUMENT;
    *out = w->heightmap->heightAt(x, z);
    return MI_OK;
}
MI_API int ora_entity_set_force(World* w, uint32_t entity, const float* f) {
    if (!w || !f || entity >= w->entities.size() || w->entities[entity].kind != MI_ENTITY_FORCE_FIELD) return MI_ERR_INVALID_ARGUMENT;
    w->entities[entity].force = vec3(f[0], f[1], f[2]);
    return MI_OK;
}
MI_API int ora_world_enable_events(World* w, uint32_t enable) {
    if (!w) return MI_ERR_INVALID_ARGUMENT;
    w->eventsEnabled = enable != 0; w->events.clear(); w->prevCollisionKeys.clear(); w->prevTriggerOverlaps.clear();
    w->prevPairColor.clear();   // the product keeps ONE history table for colours and events; enabling events restarts it
    return MI_OK;
}
MI_API int ora_world_poll_events(World* w, mi_event* out, uint32_t cap, uint32_t* count) {
    if (!w || !count) return MI_ERR_INVALID_ARGUMENT;
    *count = (uint32_t)w->events.size();
    if (!out) return MI_OK;
    if (cap < w->events.size()) return MI_ERR_CAPACITY;
    std::memcpy(out, w->events.data(), w->events.size() * sizeof(mi_event));
    w->events.clear();
    return MI_OK;
}
MI_API int ora_world_get_body_states(World* w, uint32_t n, const uint32_t* ents, float* out) {
    for (uint32_t i = 0; i < n; ++i) {
        if (ents[i] >= w->entities.size() || w->entities[ents[i]].rb < 0) return MI_ERR_INVALID_ARGUMENT;
        const RigidBody& b = w->bodies[w->entities[ents[i]].rb];
        float* o = out + 13 * (size_t)i;
        o[0] = b.p1.x; o[1] = b.p1.y; o[2] = b.p1.z; o[3] = b.r1.x; o[4] = b.r1.y; o[5] = b.r1.z; o[6] = b.r1.w;
        o[7] = b.linearVelocity.x; o[8] = b.linearVelocity.y; o[9] = b.linearVelocity.z;
        o[10] = b.angularVelocity.x; o[11] = b.angularVelocity.y; o[12] = b.angularVelocity.z;
    }
    return MI_OK;
}
MI_API int ora_world_set_body_states(World* w, uint32_t n, const uint32_t* ents, const float* in) {
    for (uint32_t i = 0; i < n; ++i) {
        if (ents[i] >= w->entities.size() || w->entities[ents[i]].rb < 0) return MI_ERR_INVALID_ARGUMENT;
        RigidBody& b = w->bodies[w->entities[ents[i]].rb];
        const float* s = in + 13 * (size_t)i;
        b.p1 = vec3(s[0], s[1], s[2]); b.r1 = quat(s[3], s[4], s[5], s[6]);
        b.linearVelocity = vec3(s[7], s[8], s[9]); b.angularVelocity = vec3(s[10], s[11], s[12]);
        b.shardKnown = 1;   // (sharded world) the caller's state is authoritative
    }
    return MI_OK;
}
// Stage dumps for bisecting mismatches.
MI_API int ora_world_get_aabbs(World* w, float* out6, uint32_t cap) {
    if (cap < w->aabbs.size()) return MI_ERR_CAPACITY;
    for (size_t i = 0; i < w->aabbs.size(); ++i) {
        out6[6 * i] = w->aabbs[i].mn.x; out6[6 * i + 1] = w->aabbs[i].mn.y; out6[6 * i + 2] = w->aabbs[i].mn.z;
        out6[6 * i + 3] = w->aabbs[i].mx.x; out6[6 * i + 4] = w->aabbs[i].mx.y; out6[6 * i + 5] = w->aabbs[i].mx.z;
    }
    return MI_OK;
}
MI_API int ora_world_get_broadphase_pairs(World* w, uint32_t* out2, uint32_t cap, uint32_t* count) {
    *count = (uint32_t)w->bpPairs.size();
    if (!out2) return MI_OK;
    if (cap < *count) return MI_ERR_CAPACITY;
    for (size_t i = 0; i < w->bpPairs.size(); ++i) { out2[2 * i] = w->bpPairs[i].a; out2[2 * i + 1] = w->bpPairs[i].b; }
    return MI_OK;
}
MI_API int ora_world_get_manifold_colors(World* w, uint32_t* out, uint32_t cap) {
    if (cap < w->manifoldColor.size()) return MI_ERR_CAPACITY;
    for (size_t i = 0; i < w->manifoldColor.size(); ++i) out[i] = w->manifoldColor[i];
    return MI_OK;
}
// Loads a checkpoint blob written by the PRODUCT (mi_world_save_checkpoint, d3d12renderer_amd/csrc/world.hip: header, entity
// transforms, body states, colour history as (key + 1, colour) pairs, trigger overlaps, constraint PODs in pool order, cloths)
// into an oracle world built from the same scene, so that the GPU can settle a full-size scene (where the oracle would take
// minutes) and both continue from the same state.  Cloth sections are not supported (MI_ERR_UNSUPPORTED).
extern "C++" {
namespace {
struct CheckpointHeader { uint32_t magic, version, numEntities, numBodies, numColliders, numHistory, numTriggerOverlaps, sapAxis; float timer; uint32_t eventsEnabled, jointCounts[6], reserved; };
template <class T> bool take(const uint8_t*& p, const uint8_t* end, T* out, size_t n) { if ((size_t)(end - p) < n * sizeof(T)) return false; std::memcpy(out, p, n * sizeof(T)); p += n * sizeof(T); return true; }
}
}
// The product's checkpoint format (csrc/world.hip "checkpoint / resume"), written from the oracle's state: lets the CPU tests save, run on,
// restore and compare without a GPU — also rank by rank in a sharded world (shard section) — and hand oracle states to the product.
MI_API int ora_world_save_checkpoint(World* w, void* out, uint64_t capacity, uint64_t* out_size) {
    if (!w || !out_size) return MI_ERR_INVALID_ARGUMENT;
    if (!w->cloths.empty()) return MI_ERR_UNSUPPORTED;
    if (w->dirtyProps) w->recalculateProperties();
    std::vector<uint8_t> blob;
    auto put = [&](const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); blob.insert(blob.end(), b, b + n); };
    CheckpointHeader h{};
    h.magic = 0x4350494Du; h.version = 1; h.numEntities = (uint32_t)w->entities.size(); h.numBodies = (uint32_t)w->bodies.size(); h.numColliders = (uint32_t)w->colliders.size();
    h.numHistory = (uint32_t)w->prevPairColor.size(); h.numTriggerOverlaps = (uint32_t)w->prevTriggerOverlaps.size();
    h.sapAxis = w->sortingAxis; h.timer = w->timer; h.eventsEnabled = w->eventsEnabled ? 1u : 0u; h.reserved = w->shard.enabled ? 1u : 0u;
    std::vector<uint8_t> pods; ora::jointsSavePods(*w, pods, h.jointCounts);
    put(&h, sizeof(h));
    for (const Entity& e : w->entities) { const float v[7] = {e.position.x, e.position.y, e.position.z, e.rotation.x, e.rotation.y, e.rotation.z, e.rotation.w}; put(v, sizeof(v)); }
    for (const RigidBody& b : w->bodies) {
        const float v[28] = {b.p0.x, b.p0.y, b.p0.z, b.r0.x, b.r0.y, b.r0.z, b.r0.w, b.p1.x, b.p1.y, b.p1.z, b.r1.x, b.r1.y, b.r1.z, b.r1.w,
                             b.linearVelocity.x, b.linearVelocity.y, b.linearVelocity.z, b.angularVelocity.x, b.angularVelocity.y, b.angularVelocity.z,
                             b.forceAccumulator.x, b.forceAccumulator.y, b.forceAccumulator.z, b.torqueAccumulator.x, b.torqueAccumulator.y, b.torqueAccumulator.z, 0.f, 0.f};
        put(v, 26 * sizeof(float));
    }
    std::vector<std::pair<uint64_t, uint32_t>> hist(w->prevPairColor.begin(), w->prevPairColor.end());
    std::sort(hist.begin(), hist.end());
    for (const auto& kv : hist) { const unsigned long long k = kv.first + 1ull; put(&k, sizeof(k)); }   // the device table stores key + 1 (0 = empty slot)
    for (const auto& kv : hist) put(&kv.second, sizeof(uint32_t));
    if (!w->prevTriggerOverlaps.empty()) put(w->prevTriggerOverlaps.data(), w->prevTriggerOverlaps.size() * sizeof(uint64_t));
    if (!pods.empty()) put(pods.data(), pods.size());
    const uint32_t numCloths = 0; put(&numCloths, sizeof(numCloths));
    if (w->shard.enabled) {
        const World::Shard& sh = w->shard; const uint32_t nb = (uint32_t)w->bodies.size();
        const uint32_t hdr[6] = {sh.desc.num_ranks, sh.desc.rank, sh.desc.tiles_x, sh.desc.tiles_z, sh.bordersPending ? 1u : 0u, (nb + 3u) & ~3u};
        put(hdr, sizeof(hdr));
        std::vector<uint8_t> known(hdr[5], 0); for (uint32_t i = 0; i < nb; ++i) known[i] = w->bodies[i].shardKnown;
        put(known.data(), known.size());
        const std::vector<float>& nx = sh.bordersPending ? sh.nextX : sh.bordersX; const std::vector<float>& nz = sh.bordersPending ? sh.nextZ : sh.bordersZ;
        if (!sh.bordersX.empty()) { put(sh.bordersX.data(), sh.bordersX.size() * sizeof(float)); put(nx.data(), nx.size() * sizeof(float)); }
        if (!sh.bordersZ.empty()) { put(sh.bordersZ.data(), sh.bordersZ.size() * sizeof(float)); put(nz.data(), nz.size() * sizeof(float)); }
    }
    *out_size = blob.size();
    if (!out) return MI_OK;
    if (capacity < blob.size()) return MI_ERR_CAPACITY;
    std::memcpy(out, blob.data(), blob.size());
    return MI_OK;
}
MI_API int ora_world_load_checkpoint(World* w, const void* data, uint64_t size) {
    if (!w || !data) return MI_ERR_INVALID_ARGUMENT;
    const uint8_t* p = static_cast<const uint8_t*>(data); const uint8_t* end = p + size;
    CheckpointHeader h;
    if (!take(p, end, &h, 1) || h.magic != 0x4350494Du || h.version != 1) return MI_ERR_INVALID_ARGUMENT;
    if (h.numEntities != w->entities.size() || h.numBodies != w->bodies.size() || h.numColliders != w->colliders.size()) return MI_ERR_INVALID_ARGUMENT;
    if (w->dirtyProps) w->recalculateProperties();
    struct F3 { float x, y, z; }; struct F4 { float x, y, z, w; };
    bool okay = true;
    for (Entity& e : w->entities) { F3 pos; F4 rot; okay = okay && take(p, end, &pos, 1) && take(p, end, &rot, 1); if (okay) { e.position = vec3(pos.x, pos.y, pos.z); e.rotation = quat(rot.x, rot.y, rot.z, rot.w); } }
    for (RigidBody& b : w->bodies) {
        F3 p0, p1, lv, av, f, t; F4 r0, r1;
        okay = okay && take(p, end, &p0, 1) && take(p, end, &r0, 1) && take(p, end, &p1, 1) && take(p, end, &r1, 1) && take(p, end, &lv, 1) && take(p, end, &av, 1) && take(p, end, &f, 1) && take(p, end, &t, 1);
        if (!okay) break;
        b.p0 = vec3(p0.x, p0.y, p0.z); b.r0 = quat(r0.x, r0.y, r0.z, r0.w); b.p1 = vec3(p1.x, p1.y, p1.z); b.r1 = quat(r1.x, r1.y, r1.z, r1.w);
        b.linearVelocity = vec3(lv.x, lv.y, lv.z); b.angularVelocity = vec3(av.x, av.y, av.z); b.forceAccumulator = vec3(f.x, f.y, f.z); b.torqueAccumulator = vec3(t.x, t.y, t.z);
    }
    std::vector<unsigned long long> keys(h.numHistory); std::vector<uint32_t> vals(h.numHistory);
    okay = okay && take(p, end, keys.data(), keys.size()) && take(p, end, vals.data(), vals.size());
    std::vector<uint64_t> overlaps(h.numTriggerOverlaps);
    okay = okay && take(p, end, overlaps.data(), overlaps.size());
    if (!okay) return MI_ERR_INVALID_ARGUMENT;
    int rc = jointsLoadPods(*w, p, end, h.jointCounts);
    if (rc != MI_OK) return rc;
    uint32_t numCloths = 0;
    if (!take(p, end, &numCloths, 1)) return MI_ERR_INVALID_ARGUMENT;
    if (numCloths != 0 || !w->cloths.empty()) return MI_ERR_UNSUPPORTED;
    // shard section (header.reserved bit 0): one RANK's view of a sharded world — which body copies are current, the tile borders in force / pending
    struct CheckpointShard { uint32_t numRanks, rank, tilesX, tilesZ, bordersPending, knownBytes; } sh{};
    std::vector<uint8_t> known; std::vector<float> curX, nextX, curZ, nextZ;
    const bool hasShard = (h.reserved & 1u) != 0u;
    if (h.reserved & ~1u) return MI_ERR_INVALID_ARGUMENT;
    if (hasShard) {
        if (!w->shard.enabled || !take(p, end, &sh, 1)) return MI_ERR_INVALID_ARGUMENT;
        const mi_shard_desc& d = w->shard.desc;
        if (sh.numRanks != d.num_ranks || sh.rank != d.rank || sh.tilesX != d.tiles_x || sh.tilesZ != d.tiles_z || sh.knownBytes != ((h.numBodies + 3u) & ~3u)) return MI_ERR_INVALID_ARGUMENT;
        known.resize(sh.knownBytes); curX.resize(w->shard.bordersX.size()); nextX.resize(curX.size()); curZ.resize(w->shard.bordersZ.size()); nextZ.resize(curZ.size());
        bool ok2 = take(p, end, known.data(), known.size());
        if (!curX.empty()) ok2 = ok2 && take(p, end, curX.data(), curX.size()) && take(p, end, nextX.data(), nextX.size());
        if (!curZ.empty()) ok2 = ok2 && take(p, end, curZ.data(), curZ.size()) && take(p, end, nextZ.data(), nextZ.size());
        if (!ok2) return MI_ERR_INVALID_ARGUMENT;
    }
    if (p != end) return MI_ERR_INVALID_ARGUMENT;
    if (w->shard.enabled) {
        for (size_t i = 0; i < w->bodies.size(); ++i) w->bodies[i].shardKnown = hasShard ? (known[i] ? 1 : 0) : 1;
        if (hasShard) { w->shard.bordersX = curX; w->shard.bordersZ = curZ; w->shard.bordersPending = sh.bordersPending != 0u; if (w->shard.bordersPending) { w->shard.nextX = nextX; w->shard.nextZ = nextZ; } }
    }
    w->timer = h.timer; w->sortingAxis = h.sapAxis; w->eventsEnabled = h.eventsEnabled != 0; w->events.clear();
    w->prevTriggerOverlaps.assign(overlaps.begin(), overlaps.end());
    w->prevPairColor.clear(); w->prevCollisionKeys.clear();
    for (uint32_t i = 0; i < h.numHistory; ++i) {
        const uint64_t key = keys[i] - 1ull;                       // the device table stores key + 1 (0 = empty slot)
        w->prevPairColor[key] = vals[i];
        if (w->eventsEnabled && (key & ((1ull << 26) - 1ull)) < kHeightmapVirtualBase) w->prevCollisionKeys.push_back(key);
    }
    std::sort(w->prevCollisionKeys.begin(), w->prevCollisionKeys.end());
    return MI_OK;
}
MI_API int ora_shard_tile_of_rank(uint32_t tx, uint32_t tz, uint32_t rank, uint32_t* out) { if (!out || !tx || !tz || rank >= tx * tz) return MI_ERR_INVALID_ARGUMENT; *out = ora::tilesInRankOrder(tx, tz)[rank]; return MI_OK; }
MI_API int ora_shard_rank_of_tile(uint32_t tx, uint32_t tz, uint32_t tile, uint32_t* out) {
    if (!out || !tx || !tz || tile >= tx * tz) return MI_ERR_INVALID_ARGUMENT;
    auto t = ora::tilesInRankOrder(tx, tz); *out = (uint32_t)(std::find(t.begin(), t.end(), tile) - t.begin()); return MI_OK;
}
MI_API int ora_world_shard_enable(World* w, const mi_shard_desc* d) {
    if (!w || !d || !d->tiles_x || !d->tiles_z || d->num_ranks != d->tiles_x * d->tiles_z || d->rank >= d->num_ranks) return MI_ERR_INVALID_ARGUMENT;
    if (!(d->tile_size_x > 0.f) || !(d->tile_size_z > 0.f) || !(d->ghost_margin > 0.f) || d->ghost_margin >= d->tile_size_x || d->ghost_margin >= d->tile_size_z) return MI_ERR_INVALID_ARGUMENT;
    if (w->orderMode != 1) return MI_ERR_UNSUPPORTED;   // (terrain is static and replicated; cloths do not interact with bodies: every rank steps all of them identically)
    World::Shard& sh = w->shard;
    sh.desc = *d;
    auto order = ora::tilesInRankOrder(d->tiles_x, d->tiles_z);
    sh.myTile = order[d->rank]; sh.peers.clear(); sh.peerRanks.clear();
    const int mx = (int)(sh.myTile % d->tiles_x), mz = (int)(sh.myTile / d->tiles_x);
    for (int z = mz - 1; z <= mz + 1; ++z) for (int x = mx - 1; x <= mx + 1; ++x) {
        if ((x == mx && z == mz) || x < 0 || z < 0 || x >= (int)d->tiles_x || z >= (int)d->tiles_z) continue;
        const uint32_t t = (uint32_t)z * d->tiles_x + (uint32_t)x;
        sh.peers.push_back(t); sh.peerRanks.push_back((uint32_t)(std::find(order.begin(), order.end(), t) - order.begin()));
    }
    sh.capacity = d->max_records ? d->max_records : std::max<uint32_t>(4096u, (uint32_t)w->bodies.size() / d->num_ranks / 4u);
    sh.sendBuf.assign(sh.peers.size(), std::vector<float>((size_t)(sh.capacity + 1u) * MI_SHARD_RECORD_FLOATS, 0.f));
    sh.bordersX.clear(); sh.bordersZ.clear(); sh.bordersPending = false;
    for (uint32_t i = 1; i < d->tiles_x; ++i) sh.bordersX.push_back((float)((double)d->origin_x + (double)i * (double)d->tile_size_x));
    for (uint32_t i = 1; i < d->tiles_z; ++i) sh.bordersZ.push_back((float)((double)d->origin_z + (double)i * (double)d->tile_size_z));
    for (RigidBody& b : w->bodies) b.shardKnown = 1;             // every rank was given the same scene
    sh.enabled = true;
    return MI_OK;
}
// Load balance (include/mi_shard.h): new interior borders, in force after the NEXT internal step's exchange (that step still simulates under the
// old ones; its messages carry what the neighbours need under the new ones).  Null = that axis stays.
MI_API int ora_world_shard_set_borders(World* w, const float* bx, const float* bz) {
    if (!w || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    World::Shard& sh = w->shard;
    const float m = sh.desc.ghost_margin;
    if (bx && !ora::shardBordersValid(sh.bordersX, bx, (uint32_t)sh.bordersX.size(), m)) return MI_ERR_INVALID_ARGUMENT;
    if (bz && !ora::shardBordersValid(sh.bordersZ, bz, (uint32_t)sh.bordersZ.size(), m)) return MI_ERR_INVALID_ARGUMENT;
    sh.nextX = bx ? std::vector<float>(bx, bx + sh.bordersX.size()) : sh.bordersX;
    sh.nextZ = bz ? std::vector<float>(bz, bz + sh.bordersZ.size()) : sh.bordersZ;
    sh.bordersPending = true;
    return MI_OK;
}
MI_API int ora_world_shard_get_borders(World* w, float* bx, float* bz) {
    if (!w || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    if (bx) std::copy(w->shard.bordersX.begin(), w->shard.bordersX.end(), bx);
    if (bz) std::copy(w->shard.bordersZ.begin(), w->shard.bordersZ.end(), bz);
    return MI_OK;
}
// owned bodies of the last step per bin of [lo, hi) along x (axis 0) or z (1), by the centre their island was classified with; the end bins take what lies outside
MI_API int ora_world_shard_histogram(World* w, uint32_t axis, float lo, float hi, uint32_t bins, uint32_t* out) {
    if (!w || !out || !w->shard.enabled || axis > 1u || !bins || !(hi > lo)) return MI_ERR_INVALID_ARGUMENT;
    std::fill(out, out + bins, 0u);
    const float scale = (float)bins / (hi - lo);
    for (uint32_t i = 0; i < w->shard.active.size(); ++i) {
        if (w->shard.active[i] != 1) continue;
        const RigidBody& r = w->bodies[w->shard.root[i]];
        const vec3 c = r.p1 + r.r1 * r.localCOG;
        const float v = ((axis ? c.z : c.x) - lo) * scale;
        const int bin = v >= (float)bins ? (int)bins - 1 : v > 0.f ? (int)v : 0;
        ++out[bin];
    }
    return MI_OK;
}
MI_API int ora_shard_balance_borders(const uint64_t* hist, uint32_t bins, float lo, float hi, uint32_t tiles, const float* cur, float margin, float* out) {
    if (!hist || !bins || !(hi > lo) || !tiles || (tiles > 1 && (!cur || !out))) return MI_ERR_INVALID_ARGUMENT;
    const uint32_t n = tiles - 1u;
    if (!n) return MI_OK;
    const std::vector<float> c(cur, cur + n);
    double total = 0; for (uint32_t b = 0; b < bins; ++b) total += (double)hist[b];
    std::vector<float> nb(c);
    if (total > 0) {
        const double width = ((double)hi - (double)lo) / (double)bins;
        uint32_t b = 0; double below = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const double target = total * (double)(i + 1u) / (double)tiles;
            while (b + 1u < bins && below + (double)hist[b] < target) { below += (double)hist[b]; ++b; }
            const double frac = hist[b] ? std::min(1.0, std::max(0.0, (target - below) / (double)hist[b])) : 0.5;
            double v = (double)lo + ((double)b + frac) * width;
            // what ONE change may do: stay between the old neighbours' borders (margin inside), keep tiles wider than the margin (x 1.25: room to move next time)
            if (i > 0) v = std::max(v, std::max((double)c[i - 1] + (double)margin, (double)nb[i - 1] + 1.25 * (double)margin));
            if (i + 1u < n) v = std::min(v, (double)c[i + 1] - (double)margin);
            // ... and move by at most two margins: the hand-over rides in ONE neighbour message, whose capacity is sized in margin strips
            v = std::min(std::max(v, (double)c[i] - 2.0 * (double)margin), (double)c[i] + 2.0 * (double)margin);
            if (i > 0) v = std::max(v, (double)nb[i - 1] + 1.25 * (double)margin);
            nb[i] = (float)v;
        }
    }
    const bool ok = ora::shardBordersValid(c, nb.data(), n, margin);
    for (uint32_t i = 0; i < n; ++i) out[i] = ok ? nb[i] : c[i];
    return MI_OK;
}
MI_API int ora_world_shard_neighbours(World* w, uint32_t* out, uint32_t* count) {
    if (!w || !count || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    *count = (uint32_t)w->shard.peers.size();
    if (out) for (size_t k = 0; k < w->shard.peers.size(); ++k) out[k] = w->shard.peerRanks[k];
    return MI_OK;
}
MI_API int ora_world_shard_counts(World* w, uint32_t* b, uint32_t* m, uint32_t* c) {
    if (!w || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    if (b) *b = w->shard.owned[0]; if (m) *m = w->shard.owned[1]; if (c) *c = w->shard.owned[2];
    return MI_OK;
}
MI_API int ora_world_shard_owned_entities(World* w, uint32_t* out, uint32_t cap, uint32_t* count) {
    if (!w || !count || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    uint32_t n = 0;
    for (uint32_t i = 0; i < w->shard.active.size(); ++i) if (w->shard.active[i] == 1) { if (out && n < cap) out[n] = w->bodies[i].entity; ++n; }
    *count = n;
    return (out && n > cap) ? MI_ERR_CAPACITY : MI_OK;
}
MI_API int ora_world_shard_message_bytes(World* w, uint64_t* out) {
    if (!w || !out || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    *out = (uint64_t)(w->shard.capacity + 1u) * MI_SHARD_RECORD_FLOATS * sizeof(float); return MI_OK;
}
MI_API int ora_world_shard_export(World* w, uint32_t slot, void* out) {
    if (!w || !out || !w->shard.enabled || slot >= w->shard.sendBuf.size()) return MI_ERR_INVALID_ARGUMENT;
    std::memcpy(out, w->shard.sendBuf[slot].data(), w->shard.sendBuf[slot].size() * sizeof(float)); return MI_OK;
}
MI_API int ora_world_shard_import(World* w, const void* msg) {
    if (!w || !msg || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    uint32_t count; std::memcpy(&count, msg, 4);
    if (count > w->shard.capacity) return MI_ERR_CAPACITY;
    const float* f = static_cast<const float*>(msg);
    for (uint32_t r = 0; r < count; ++r) {
        const float* s = f + (size_t)(r + 1u) * MI_SHARD_RECORD_FLOATS;
        uint32_t b; std::memcpy(&b, s, 4);
        if (b >= w->bodies.size()) continue;
        RigidBody& rb = w->bodies[b];
        rb.p1 = vec3(s[1], s[2], s[3]); rb.r1 = quat(s[4], s[5], s[6], s[7]);
        rb.linearVelocity = vec3(s[8], s[9], s[10]); rb.angularVelocity = vec3(s[11], s[12], s[13]);
        rb.shardKnown = 1;
    }
    return MI_OK;
}
// ---- exact seam (include/mi_shard.h)
MI_API int ora_world_set_seam_tiling(World* w, const mi_shard_desc* d) {
    if (!w) return MI_ERR_INVALID_ARGUMENT;
    w->prevPairColor.clear();
    if (!d) { w->seam.tiling = false; return MI_OK; }
    if (w->shard.enabled || w->orderMode != 1) return MI_ERR_UNSUPPORTED;
    if (!d->tiles_x || !d->tiles_z || !(d->tile_size_x > 0.f) || !(d->tile_size_z > 0.f) || !(d->ghost_margin > 0.f) || 2.f * d->ghost_margin > d->tile_size_x || 2.f * d->ghost_margin > d->tile_size_z) return MI_ERR_INVALID_ARGUMENT;
    w->seam.bx.clear(); w->seam.bz.clear(); w->seam.margin = d->ghost_margin;
    for (uint32_t i = 1; i < d->tiles_x; ++i) w->seam.bx.push_back((float)((double)d->origin_x + (double)i * (double)d->tile_size_x));
    for (uint32_t i = 1; i < d->tiles_z; ++i) w->seam.bz.push_back((float)((double)d->origin_z + (double)i * (double)d->tile_size_z));
    w->seam.tiling = true;
    return MI_OK;
}
MI_API int ora_world_shard_set_exact_seam(World* w, uint32_t enable, int (*fn)(void*, World*, uint32_t), void* user) {
    if (!w || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    if (enable && w->shard.desc.tiles_x > 1u && w->shard.desc.tiles_z > 1u) return MI_ERR_UNSUPPORTED;   // slabs only: at a corner a shared body is seen by four tiles
    if (enable && (2.f * w->shard.desc.ghost_margin > w->shard.desc.tile_size_x || 2.f * w->shard.desc.ghost_margin > w->shard.desc.tile_size_z)) return MI_ERR_INVALID_ARGUMENT;   // a body is shared across ONE border
    if (w->seam.exact != (enable != 0u)) w->prevPairColor.clear();   // the colour ranges mean something else from here on
    w->seam.exact = enable != 0u; w->seam.fn = fn; w->seam.user = user; w->seam.error = MI_OK;
    return MI_OK;
}
MI_API int ora_world_shard_sweep_message_bytes(World* w, uint64_t* out) {
    if (!w || !out || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    *out = (uint64_t)(w->shard.capacity + 1u) * MI_SHARD_SWEEP_FLOATS * sizeof(float); return MI_OK;
}
// inside the sweep callback: (body index, linear velocity, angular velocity) of every body this rank owns that neighbour `slot` holds as a ghost
MI_API int ora_world_shard_export_sweep(World* w, uint32_t slot, void* out) {
    if (!w || !out || !w->shard.enabled || !w->seam.exact || slot >= w->shard.peers.size()) return MI_ERR_INVALID_ARGUMENT;
    float* msg = static_cast<float*>(out);
    std::memset(msg, 0, (size_t)(w->shard.capacity + 1u) * MI_SHARD_SWEEP_FLOATS * sizeof(float));
    uint32_t n = 0;
    for (uint32_t i = 0; i < (uint32_t)w->bodies.size(); ++i) {
        if (w->shard.active[i] != 1) continue;
        const vec3 c = w->seam.cogStart[i];
        if (!ora::shardInExtended(w->shard, w->shard.bordersX, w->shard.bordersZ, w->shard.peers[slot], c.x, c.z)) continue;
        if (n < w->shard.capacity) {
            float* o = msg + (size_t)(n + 1u) * MI_SHARD_SWEEP_FLOATS;
            std::memcpy(o, &i, 4);
            const GlobalState& g = w->rb[i];
            o[1] = g.linearVelocity.x; o[2] = g.linearVelocity.y; o[3] = g.linearVelocity.z;
            o[4] = g.angularVelocity.x; o[5] = g.angularVelocity.y; o[6] = g.angularVelocity.z;
        }
        ++n;
    }
    std::memcpy(msg, &n, 4);
    return n > w->shard.capacity ? MI_ERR_CAPACITY : MI_OK;
}
MI_API int ora_world_shard_import_sweep(World* w, const void* msg) {
    if (!w || !msg || !w->shard.enabled || !w->seam.exact) return MI_ERR_INVALID_ARGUMENT;
    uint32_t count; std::memcpy(&count, msg, 4);
    if (count > w->shard.capacity) return MI_ERR_CAPACITY;
    const float* f = static_cast<const float*>(msg);
    for (uint32_t r = 0; r < count; ++r) {
        const float* s = f + (size_t)(r + 1u) * MI_SHARD_SWEEP_FLOATS;
        uint32_t b; std::memcpy(&b, s, 4);
        if (b >= w->bodies.size() || w->shard.active[b] != 2) continue;   // only a ghost's copy is replaced
        w->rb[b].linearVelocity = vec3(s[1], s[2], s[3]); w->rb[b].angularVelocity = vec3(s[4], s[5], s[6]);
    }
    return MI_OK;
}
MI_API int ora_world_seam_stats(World* w, uint32_t* manifolds, uint32_t* colors, uint32_t* violations) {
    if (!w) return MI_ERR_INVALID_ARGUMENT;
    if (manifolds) *manifolds = w->seam.manifolds;
    if (colors) *colors = w->seam.colors;
    if (violations) *violations = w->seam.violations;
    return MI_OK;
}
// Global sweep axis of a sharded world: every rank's sums (the colliders it owns; rank 0 also the ones without a rigid body) added over
// the ranks by the caller, handed back to every rank — the axis of the next step is then the single world's, whatever the tiling.
MI_API int ora_world_shard_axis_sums(World* w, uint64_t* out9) {
    if (!w || !out9 || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    std::memcpy(out9, w->axisSums, sizeof(w->axisSums)); return MI_OK;
}
MI_API int ora_world_shard_set_axis_sums(World* w, const uint64_t* global9) {
    if (!w || !global9 || !w->shard.enabled) return MI_ERR_INVALID_ARGUMENT;
    w->sortingAxis = ora::axisFromSums(global9, (uint32_t)w->colliders.size()); return MI_OK;
}
MI_API int ora_debug_set_solve_order(World* w, const uint32_t* pairs, uint32_t count) {   // (mi_debug_set_solve_order)
    if (!w || (count && !pairs) || w->orderMode != 1) return MI_ERR_INVALID_ARGUMENT;
    w->debugOrder.resize(count);
    for (uint32_t i = 0; i < count; ++i) w->debugOrder[i] = Pair{pairs[2 * i], pairs[2 * i + 1]};
    w->debugOrderPending = true; w->debugOrderError = false;
    return MI_OK;
}
MI_API int ora_debug_set_sweep_axis(World* w, uint32_t axis) { if (!w || axis > 2u) return MI_ERR_INVALID_ARGUMENT; w->sortingAxis = axis; return MI_OK; }   // (mi_debug_set_sweep_axis)
MI_API uint32_t ora_axis_from_sums(const uint64_t* sums9, uint32_t numColliders) { return ora::axisFromSums(sums9, numColliders); }
MI_API float ora_det_atan2f(float y, float x) { return det_atan2f(y, x); }
MI_API float ora_det_acosf(float x) { return det_acosf(x); }
MI_API float ora_det_sinf(float x) { return det_sinf(x); }
MI_API float ora_det_cosf(float x) { return det_cosf(x); }

}  // extern "C"
