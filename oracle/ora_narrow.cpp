// ora_narrow.cpp — TEST INFRASTRUCTURE (CPU oracle), see ora_world.h header note.
// Scalar shape-pair intersection() routines of src/physics/collision_narrow.cpp, same operation
// order.  Normal points from A to B; penetrationDepth >= 0; contact point = midpoint.
#include "ora_world.h"
#include <cstring>
#include <algorithm>

namespace ora {

struct VPP { vec3 vertex; float penetrationDepth; };          // vertex_penetration_pair (collision_narrow.cpp:50-54)
struct ClipPoly { VPP points[16]; uint32_t numPoints = 0; };  // clipping_polygon (148-152)

static void setContact(ContactManifold& m, uint32_t i, vec3 p, float d) { m.points[i] = p; m.depths[i] = d; }

// findStableContactManifold — collision_narrow.cpp:56-146
static void findStableContactManifold(VPP* v, uint32_t n, vec3 normal, ContactManifold& out) {
    if (n > 4) {
        vec3 searchDir = getTangent(normal);
        float bestDistance = dot(searchDir, v[0].vertex);
        uint32_t ri = 0;
        for (uint32_t i = 1; i < n; ++i) { float d = dot(searchDir, v[i].vertex); if (d > bestDistance) { ri = i; bestDistance = d; } }
        setContact(out, 0, v[ri].vertex, v[ri].penetrationDepth);
        bestDistance = 0.f; ri = 0;
        for (uint32_t i = 0; i < n; ++i) { float sq = squaredLength(v[i].vertex - out.points[0]); if (sq > bestDistance) { ri = i; bestDistance = sq; } }
        setContact(out, 1, v[ri].vertex, v[ri].penetrationDepth);
        float bestArea = 0.f; ri = 0;
        for (uint32_t i = 0; i < n; ++i) {
            vec3 qa = out.points[0] - v[i].vertex, qb = out.points[1] - v[i].vertex;
            float area = 0.5f * dot(cross(qa, qb), normal);
            if (area > bestArea) { ri = i; bestArea = area; }
        }
        setContact(out, 2, v[ri].vertex, v[ri].penetrationDepth);
        bestArea = 0.f; ri = 0;
        for (uint32_t i = 0; i < n; ++i) {
            vec3 qa = out.points[0] - v[i].vertex, qb = out.points[1] - v[i].vertex, qc = out.points[2] - v[i].vertex;
            float area1 = 0.5f * dot(cross(qa, qb), normal);
            float area2 = 0.5f * dot(cross(qb, qc), normal);
            float area3 = 0.5f * dot(cross(qc, qa), normal);
            float area = fmax2(fmax2(area1, area2), area3);
            if (area > bestArea) { ri = i; bestArea = area; }
        }
        setContact(out, 3, v[ri].vertex, v[ri].penetrationDepth);
        out.numContacts = 4;
    } else {
        out.numContacts = n;
        for (uint32_t i = 0; i < n; ++i) setContact(out, i, v[i].vertex, v[i].penetrationDepth);
    }
}

// clipAgainstPlane — 154-163
static VPP clipAgainstPlane(VPP a, VPP b, float aDist, float bDist) {
    aDist = std::fabs(aDist); bDist = std::fabs(bDist);
    float total = aDist + bDist;
    float t = aDist / total;
    return VPP{lerp(a.vertex, b.vertex, t), lerpf(a.penetrationDepth, b.penetrationDepth, t)};
}

// sutherlandHodgmanClipping — 166-222 (planes point inside)
static void sutherlandHodgmanClipping(ClipPoly& input, const vec4* planes, uint32_t numPlanes, ClipPoly& output) {
    ClipPoly* in = &input; ClipPoly* out = &output;
    uint32_t clipIndex = 0;
    for (; clipIndex < numPlanes; ++clipIndex) {
        vec4 pl = planes[clipIndex];
        out->numPoints = 0;
        if (in->numPoints == 0) break;
        VPP startPoint = in->points[in->numPoints - 1];
        for (uint32_t i = 0; i < in->numPoints; ++i) {
            VPP endPoint = in->points[i];
            float startDist = signedDistanceToPlane(startPoint.vertex, pl);
            float endDist = signedDistanceToPlane(endPoint.vertex, pl);
            bool startInside = startDist > 0.f, endInside = endDist > 0.f;
            if (startInside && endInside) out->points[out->numPoints++] = endPoint;
            else if (startInside) out->points[out->numPoints++] = clipAgainstPlane(startPoint, endPoint, startDist, endDist);
            else if (!startInside && endInside) {
                out->points[out->numPoints++] = clipAgainstPlane(startPoint, endPoint, startDist, endDist);
                out->points[out->numPoints++] = endPoint;
            }
            startPoint = endPoint;
        }
        ClipPoly* tmp = in; in = out; out = tmp;
    }
    if (clipIndex % 2 == 0) {
        for (uint32_t i = 0; i < input.numPoints; ++i) output.points[i] = input.points[i];
        output.numPoints = input.numPoints;
    }
}

static uint32_t maxElementIndex(vec3 p) { return (p.x > p.y) ? ((p.x > p.z) ? 0 : 2) : ((p.y > p.z) ? 1 : 2); }

// getAABBClippingPlanes — 225-256
static void getAABBClippingPlanes(vec3 radius, vec3 normal, vec3* pts, vec3* nrm) {
    vec3 p = vabs(normal);
    uint32_t me = maxElementIndex(p);
    uint32_t axis0 = (me + 1) % 3, axis1 = (me + 2) % 3;
    { vec3 n(0.f); n[axis0] = 1.f; nrm[0] = n; pts[0] = -radius; }
    { vec3 n(0.f); n[axis1] = 1.f; nrm[1] = n; pts[1] = -radius; }
    { vec3 n(0.f); n[axis0] = -1.f; nrm[2] = n; pts[2] = radius; }
    { vec3 n(0.f); n[axis1] = -1.f; nrm[3] = n; pts[3] = radius; }
}

// getAABBIncidentVertices — 259-293
static void getAABBIncidentVertices(vec3 radius, vec3 normal, ClipPoly& poly) {
    vec3 p = vabs(normal);
    uint32_t me = maxElementIndex(p);
    float s = normal[me] < 0.f ? 1.f : -1.f;
    uint32_t axis0 = (me + 1) % 3, axis1 = (me + 2) % 3;
    float d = radius[me] * s;
    float min0 = -radius[axis0], min1 = -radius[axis1], max0 = radius[axis0], max1 = radius[axis1];
    poly.numPoints = 4;
    poly.points[0].vertex[me] = d; poly.points[0].vertex[axis0] = min0; poly.points[0].vertex[axis1] = min1;
    poly.points[1].vertex[me] = d; poly.points[1].vertex[axis0] = max0; poly.points[1].vertex[axis1] = min1;
    poly.points[2].vertex[me] = d; poly.points[2].vertex[axis0] = max0; poly.points[2].vertex[axis1] = max1;
    poly.points[3].vertex[me] = d; poly.points[3].vertex[axis0] = min0; poly.points[3].vertex[axis1] = max1;
}

// getAABBReferencePlane — 295-303
static vec4 getAABBReferencePlane(vec3 mn, vec3 mx, vec3 normal) {
    vec3 point((normal.x < 0.f) ? mn.x : mx.x, (normal.y < 0.f) ? mn.y : mx.y, (normal.z < 0.f) ? mn.z : mx.z);
    return createPlane(point, normal);
}

// getAABBIncidentEdge — 305-337
static void getAABBIncidentEdge(vec3 r, vec3 normal, vec3& outA, vec3& outB) {
    vec3 p = vabs(normal);
    outA = vec3(r.x, r.y, r.z);
    if (p.x > p.y) {
        if (p.y > p.z) outB = vec3(r.x, r.y, -r.z); else outB = vec3(r.x, -r.y, r.z);
    } else {
        if (p.x > p.z) outB = vec3(r.x, r.y, -r.z); else outB = vec3(-r.x, r.y, r.z);
    }
    float sx = normal.x < 0.f ? -1.f : 1.f, sy = normal.y < 0.f ? -1.f : 1.f, sz = normal.z < 0.f ? -1.f : 1.f;
    outA = outA * vec3(sx, sy, sz);
    outB = outB * vec3(sx, sy, sz);
}

// clipPointsAndBuildContact — 339-369
static bool clipPointsAndBuildContact(ClipPoly& poly, const vec4* planes, uint32_t numPlanes, vec4 refPlane, ContactManifold& out) {
    ClipPoly clipped;
    sutherlandHodgmanClipping(poly, planes, numPlanes, clipped);
    if (clipped.numPoints > 0) {
        vec3 rn(refPlane.x, refPlane.y, refPlane.z);
        for (uint32_t i = 0; i < clipped.numPoints; ++i) {
            if (clipped.points[i].penetrationDepth < 0.f) {
                clipped.points[i] = clipped.points[clipped.numPoints - 1];
                --clipped.numPoints;
                --i;
            } else {
                clipped.points[i].vertex += rn * clipped.points[i].penetrationDepth;
            }
        }
        if (clipped.numPoints > 0) {
            findStableContactManifold(clipped.points, clipped.numPoints, out.normal, out);
            return true;
        }
    }
    return false;
}

// closestPoint_PointSegment — bounding_volumes.h:365-371
vec3 closestPoint_PointSegment(vec3 q, vec3 la, vec3 lb) {
    vec3 ab = lb - la;
    float t = dot(q - la, ab) / squaredLength(ab);
    t = clampf(t, 0.f, 1.f);
    return la + t * ab;
}
// closestPoint_PointAABB — bounding_volumes.h:373-384
static vec3 closestPoint_PointAABB(vec3 q, vec3 mn, vec3 mx) {
    vec3 r;
    for (int i = 0; i < 3; ++i) { float v = q[i]; if (v < mn[i]) v = mn[i]; if (v > mx[i]) v = mx[i]; r[i] = v; }
    return r;
}
// closestPoint_SegmentSegment — bounding_volumes.cpp:1251-1315
float closestPoint_SegmentSegment(vec3 l1a, vec3 l1b, vec3 l2a, vec3 l2b, vec3& c1, vec3& c2) {
    float s, t;
    vec3 d1 = l1b - l1a, d2 = l2b - l2a, r = l1a - l2a;
    float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    if (a <= kEps && e <= kEps) { c1 = l1a; c2 = l2a; return dot(c1 - c2, c1 - c2); }
    if (a <= kEps) { s = 0.f; t = f / e; t = clampf(t, 0.f, 1.f); }
    else {
        float c = dot(d1, r);
        if (e <= kEps) { t = 0.f; s = clampf(-c / a, 0.f, 1.f); }
        else {
            float b = dot(d1, d2);
            float denom = a * e - b * b;
            if (denom != 0.f) s = clampf((b * f - c * e) / denom, 0.f, 1.f); else s = 0.f;
            t = (b * s + f) / e;
            if (t < 0.f) { t = 0.f; s = clampf(-c / a, 0.f, 1.f); }
            else if (t > 1.f) { t = 1.f; s = clampf((b - c) / a, 0.f, 1.f); }
        }
    }
    c1 = l1a + d1 * s;
    c2 = l2a + d2 * t;
    return squaredLength(c1 - c2);
}

// ---------------------------------------------------------------- sphere tests (374-521)

static bool sphereSphere(vec3 c1, float r1, vec3 c2, float r2, ContactManifold& out) {  // 374-401
    vec3 n = c2 - c1;
    float radiusSum = r2 + r1;
    float sq = squaredLength(n);
    if (sq <= radiusSum * radiusSum) {
        float distance;
        if (sq == 0.f) { distance = 0.f; out.normal = vec3(0.f, 1.f, 0.f); }
        else { distance = std::sqrt(sq); out.normal = n / distance; }
        out.numContacts = 1;
        out.depths[0] = radiusSum - distance;
        out.points[0] = 0.5f * (c1 + r1 * out.normal + c2 - r2 * out.normal);
        return true;
    }
    return false;
}

static bool sphereCylinder(vec3 sc, float sr, vec3 ca, vec3 cb, float cr, ContactManifold& out) {  // 409-452
    vec3 ab = cb - ca;
    float t = dot(sc - ca, ab) / squaredLength(ab);
    if (t >= 0.f && t <= 1.f) return sphereSphere(sc, sr, lerp(ca, cb, t), cr, out);
    vec3 p = (t <= 0.f) ? ca : cb;
    vec3 up = (t <= 0.f) ? -ab : ab;
    vec3 projectedDirToCenter = normalize(cross(cross(up, sc - p), up));
    vec3 endA = p + projectedDirToCenter * cr;
    vec3 endB = p - projectedDirToCenter * cr;
    vec3 closestToSphere = closestPoint_PointSegment(sc, endA, endB);
    vec3 normal = closestToSphere - sc;
    float sq = squaredLength(normal);
    if (sq <= sr * sr) {
        float distance;
        if (sq == 0.f) { distance = 0.f; out.normal = -normalize(up); }
        else { distance = std::sqrt(sq); out.normal = normal / distance; }
        out.numContacts = 1;
        out.depths[0] = sr - distance;
        out.points[0] = closestToSphere + 0.5f * out.depths[0] * normal;
        return true;
    }
    return false;
}

static bool sphereAABB(vec3 sc, float sr, vec3 mn, vec3 mx, ContactManifold& out) {  // 454-481
    vec3 p = closestPoint_PointAABB(sc, mn, mx);
    vec3 n = p - sc;
    float sq = squaredLength(n);
    if (sq <= sr * sr) {
        float dist = 0.f;
        if (sq > 0.f) { dist = std::sqrt(sq); n /= dist; }
        else n = vec3(0.f, 1.f, 0.f);
        out.numContacts = 1;
        out.normal = n;
        out.depths[0] = sr - dist;
        out.points[0] = 0.5f * (p + sc + n * sr);
        return true;
    }
    return false;
}

static bool sphereOBB(vec3 sc, float sr, quat orot, vec3 oc, vec3 orad, ContactManifold& out) {  // 483-497
    vec3 mn = oc - orad, mx = oc + orad;
    vec3 sc_ = conjugate(orot) * (sc - oc) + oc;
    if (sphereAABB(sc_, sr, mn, mx, out)) {
        out.normal = orot * out.normal;
        out.points[0] = orot * (out.points[0] - oc) + oc;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------- capsule tests (523-703)

// capsule vs capsule (523-612) and capsule vs cylinder (614-703) share the parallel-case code; the
// only difference is which end-cap routine resolves the non-overlapping / skew cases.
static bool capsuleVsSegmentShape(const Shape& a, const Shape& b, bool bIsCylinder, ContactManifold& out) {
    auto endTest = [&](vec3 ac, vec3 bc) -> bool {
        if (bIsCylinder) return sphereCylinder(ac, a.radius, b.a, b.b, b.radius, out);
        return sphereSphere(ac, a.radius, bc, b.radius, out);
    };
    vec3 aDir = a.b - a.a;
    vec3 bDir = normalize(b.b - b.a);
    float aDirLength = length(aDir);
    aDir *= 1.f / aDirLength;
    float parallel = dot(aDir, bDir);
    if (std::fabs(parallel) > 0.99f) {
        vec3 pAa = a.a, pAb = a.b, pBa = b.a, pBb = b.b;
        if (parallel < 0.f) std::swap(pBa, pBb);
        vec3 referencePoint = a.a;
        float a0 = 0.f, a1 = aDirLength;
        float b0 = dot(aDir, pBa - referencePoint);
        float b1 = dot(aDir, pBb - referencePoint);
        float left = fmax2(a0, b0), right = fmin2(a1, b1);
        if (right < left) {
            if (a0 > b1) return endTest(pAa, pBb);
            else return endTest(pAb, pBa);
        }
        vec3 contactA0 = referencePoint + left * aDir;
        vec3 contactA1 = referencePoint + right * aDir;
        vec3 contactB0 = closestPoint_PointSegment(contactA0, pBa, pBb);
        vec3 contactB1 = contactB0 + (right - left) * aDir;
        vec3 normal = contactB0 - contactA0;
        float d = length(normal);
        if (d < kEps) { d = 0.f; normal = vec3(0.f, 1.f, 0.f); }
        else normal /= d;
        float radiusSum = a.radius + b.radius;
        float penetration = radiusSum - d;
        if (penetration < 0.f) return false;
        out.normal = normal;
        out.numContacts = 2;
        out.depths[0] = penetration; out.points[0] = (contactA0 + contactB0) * 0.5f;
        out.depths[1] = penetration; out.points[1] = (contactA1 + contactB1) * 0.5f;
        return true;
    } else {
        vec3 cp1, cp2;
        closestPoint_SegmentSegment(a.a, a.b, b.a, b.b, cp1, cp2);
        return endTest(cp1, cp2);
    }
}

// ---------------------------------------------------------------- AABB / OBB tests

static bool aabbAABB(vec3 amn, vec3 amx, vec3 bmn, vec3 bmx, ContactManifold& out) {  // 1074-1140
    vec3 centerA = (amn + amx) * 0.5f, centerB = (bmn + bmx) * 0.5f;
    vec3 radiusA = (amx - amn) * 0.5f, radiusB = (bmx - bmn) * 0.5f;
    vec3 d = centerB - centerA;
    vec3 p = (radiusB + radiusA) - vabs(d);
    if (p.x < 0.f || p.y < 0.f || p.z < 0.f) return false;
    uint32_t me = (p.x < p.y) ? ((p.x < p.z) ? 0 : 2) : ((p.y < p.z) ? 1 : 2);
    float s = d[me] < 0.f ? -1.f : 1.f;
    float penetration = p[me] * s;
    vec3 normal(0.f); normal[me] = s;
    out.normal = normal;
    out.numContacts = 4;
    uint32_t axis0 = (me + 1) % 3, axis1 = (me + 2) % 3;
    float min0 = fmax2(amn[axis0], bmn[axis0]), min1 = fmax2(amn[axis1], bmn[axis1]);
    float max0 = fmin2(amx[axis0], bmx[axis0]), max1 = fmin2(amx[axis1], bmx[axis1]);
    float depth = centerA[me] + radiusA[me] - penetration * 0.5f;
    float c0[4] = {min0, min0, max0, max0}, c1[4] = {min1, max1, min1, max1};
    for (int i = 0; i < 4; ++i) {
        out.depths[i] = penetration;
        vec3 pt(0.f); pt[axis0] = c0[i]; pt[axis1] = c1[i]; pt[me] = depth;
        out.points[i] = pt;
    }
    return true;
}

static vec3 obbSupport(quat rot, vec3 center, vec3 radius, vec3 dir) {  // obb_support_fn, collision_gjk.h:59-75
    dir = conjugate(rot) * dir;
    vec3 r(dir.x < 0.f ? -radius.x : radius.x, dir.y < 0.f ? -radius.y : radius.y, dir.z < 0.f ? -radius.z : radius.z);
    return center + rot * r;
}

// OBB vs OBB — 1179-1527
static bool obbOBB(quat arot, vec3 acen, vec3 arad, quat brot, vec3 bcen, vec3 brad, ContactManifold& out) {
    vec3 ax = arot * vec3(1.f, 0.f, 0.f), ay = arot * vec3(0.f, 1.f, 0.f), az = arot * vec3(0.f, 0.f, 1.f);
    vec3 bx = brot * vec3(1.f, 0.f, 0.f), by = brot * vec3(0.f, 1.f, 0.f), bz = brot * vec3(0.f, 0.f, 1.f);
    mat3 r;
    r.m00 = dot(ax, bx); r.m10 = dot(ay, bx); r.m20 = dot(az, bx);
    r.m01 = dot(ax, by); r.m11 = dot(ay, by); r.m21 = dot(az, by);
    r.m02 = dot(ax, bz); r.m12 = dot(ay, bz); r.m22 = dot(az, bz);
    vec3 tw = bcen - acen;
    vec3 t = conjugate(arot) * tw;
    bool parallel = false;
    mat3 absR;
    for (int i = 0; i < 9; ++i) {
        absR.data()[i] = std::fabs(r.data()[i]) + kEps;
        if (absR.data()[i] >= 0.99f) parallel = true;
    }
    float ra, rb;
    float minPenetration = FLT_MAX;
    vec3 normal;
    bool bFace = false;
    for (int i = 0; i < 3; ++i) {
        ra = arad[i];
        rb = dot(row(absR, i), brad);
        float d = t[i];
        float penetration = ra + rb - std::fabs(d);
        if (penetration < 0.f) return false;
        if (penetration < minPenetration) { minPenetration = penetration; normal = vec3(0.f); normal[i] = 1.f; }
    }
    for (int i = 0; i < 3; ++i) {
        ra = dot(col(absR, i), arad);
        rb = brad[i];
        float d = dot(col(r, i), t);
        float penetration = ra + rb - std::fabs(d);
        if (penetration < 0.f) return false;
        if (penetration < minPenetration) { minPenetration = penetration; normal = vec3(0.f); normal[i] = 1.f; bFace = true; }
    }
    bool edgeCollision = false;
    vec3 edgeNormal;
    if (!parallel) {
        float penetration; vec3 n; float l;
#define ORA_EDGE_TEST(RA, RB, DIST, NX, NY, NZ) \
        ra = RA; rb = RB; penetration = ra + rb - std::fabs(DIST); \
        if (penetration < 0.f) return false; \
        n = vec3(NX, NY, NZ); l = 1.f / length(n); penetration *= l; \
        if (penetration < minPenetration) { minPenetration = penetration; edgeNormal = n * l; edgeCollision = true; }
        ORA_EDGE_TEST(arad.y * absR.m20 + arad.z * absR.m10, brad.y * absR.m02 + brad.z * absR.m01, t.z * r.m10 - t.y * r.m20, 0.f, -r.m20, r.m10)  // a.x x b.x
        ORA_EDGE_TEST(arad.y * absR.m21 + arad.z * absR.m11, brad.x * absR.m02 + brad.z * absR.m00, t.z * r.m11 - t.y * r.m21, 0.f, -r.m21, r.m11)  // a.x x b.y
        ORA_EDGE_TEST(arad.y * absR.m22 + arad.z * absR.m12, brad.x * absR.m01 + brad.y * absR.m00, t.z * r.m12 - t.y * r.m22, 0.f, -r.m22, r.m12)  // a.x x b.z
        ORA_EDGE_TEST(arad.x * absR.m20 + arad.z * absR.m00, brad.y * absR.m12 + brad.z * absR.m11, t.x * r.m20 - t.z * r.m00, r.m20, 0.f, -r.m00)  // a.y x b.x
        ORA_EDGE_TEST(arad.x * absR.m21 + arad.z * absR.m01, brad.x * absR.m12 + brad.z * absR.m10, t.x * r.m21 - t.z * r.m01, r.m21, 0.f, -r.m01)  // a.y x b.y
        ORA_EDGE_TEST(arad.x * absR.m22 + arad.z * absR.m02, brad.x * absR.m11 + brad.y * absR.m10, t.x * r.m22 - t.z * r.m02, r.m22, 0.f, -r.m02)  // a.y x b.z
        ORA_EDGE_TEST(arad.x * absR.m10 + arad.y * absR.m00, brad.y * absR.m22 + brad.z * absR.m21, t.y * r.m00 - t.x * r.m10, -r.m10, r.m00, 0.f)  // a.z x b.x
        ORA_EDGE_TEST(arad.x * absR.m11 + arad.y * absR.m01, brad.x * absR.m22 + brad.z * absR.m20, t.y * r.m01 - t.x * r.m11, -r.m11, r.m01, 0.f)  // a.z x b.y
        ORA_EDGE_TEST(arad.x * absR.m12 + arad.y * absR.m02, brad.x * absR.m21 + brad.y * absR.m20, t.y * r.m02 - t.x * r.m12, -r.m12, r.m02, 0.f)  // a.z x b.z
#undef ORA_EDGE_TEST
    }
    bool faceCollision = !edgeCollision;
    if (faceCollision) { if (bFace) normal = r * normal; }
    else normal = edgeNormal;
    normal = arot * normal;
    if (dot(normal, tw) < 0.f) normal = -normal;
    out.normal = normal;
    if (faceCollision) {
        vec3 clipPts[4], clipNrm[4];
        ClipPoly poly;
        vec4 plane;
        if (!bFace) {
            getAABBClippingPlanes(arad, conjugate(arot) * normal, clipPts, clipNrm);
            getAABBIncidentVertices(brad, conjugate(brot) * normal, poly);
            for (int i = 0; i < 4; ++i) {
                clipPts[i] = arot * clipPts[i] + acen;
                clipNrm[i] = arot * clipNrm[i];
                poly.points[i].vertex = brot * poly.points[i].vertex + bcen;
            }
            vec3 refPoint = obbSupport(arot, acen, arad, normal);
            plane = createPlane(refPoint, normal);
        } else {
            getAABBClippingPlanes(brad, conjugate(brot) * -normal, clipPts, clipNrm);
            getAABBIncidentVertices(arad, conjugate(arot) * -normal, poly);
            for (int i = 0; i < 4; ++i) {
                clipPts[i] = brot * clipPts[i] + bcen;
                clipNrm[i] = brot * clipNrm[i];
                poly.points[i].vertex = arot * poly.points[i].vertex + acen;
            }
            vec3 refPoint = obbSupport(brot, bcen, brad, -normal);
            plane = createPlane(refPoint, -normal);
        }
        vec4 clipPlanes[4];
        for (int i = 0; i < 4; ++i) {
            clipPlanes[i] = createPlane(clipPts[i], clipNrm[i]);
            poly.points[i].penetrationDepth = -signedDistanceToPlane(poly.points[i].vertex, plane);
        }
        if (!clipPointsAndBuildContact(poly, clipPlanes, 4, plane, out)) return false;
    } else {
        vec3 a0, a1, b0, b1;
        getAABBIncidentEdge(arad, conjugate(arot) * normal, a0, a1);
        getAABBIncidentEdge(brad, conjugate(brot) * -normal, b0, b1);
        a0 = arot * a0 + acen; a1 = arot * a1 + acen;
        b0 = brot * b0 + bcen; b1 = brot * b1 + bcen;
        vec3 pa, pb;
        float sq = closestPoint_SegmentSegment(a0, a1, b0, b1, pa, pb);
        out.numContacts = 1;
        out.depths[0] = std::sqrt(sq);
        out.points[0] = (pa + pb) * 0.5f;
    }
    return true;
}

// ---------------------------------------------------------------- GJK + EPA based tests

static bool gjkEpaSingle(const World& w, const Shape& a, const Shape& b, ContactManifold& out, EpaResult& epa) {
    SupportShape sa{&a, a.type == T_HULL ? &w.hulls[a.hull] : nullptr};
    SupportShape sb{&b, b.type == T_HULL ? &w.hulls[b.hull] : nullptr};
    GjkSimplex simplex;
    if (!gjkIntersectionTest(sa, sb, simplex)) return false;
    epaCollisionInfo(simplex, sa, sb, epa);  // failure status ignored (collision_narrow.cpp:509-512)
    out.normal = epa.normal;
    out.numContacts = 1;
    out.depths[0] = epa.penetrationDepth;
    out.points[0] = epa.point;
    return true;
}

// capsule vs AABB — 705-768
static bool capsuleAABB(const World& w, const Shape& c, const Shape& a, ContactManifold& out) {
    EpaResult epa;
    if (!gjkEpaSingle(w, c, a, out, epa)) return false;
    vec3 normal = epa.normal;
    if (std::fabs(normal.x) > 0.99f || std::fabs(normal.y) > 0.99f || std::fabs(normal.z) > 0.99f) {
        vec3 axis = normalize(c.b - c.a);
        if (std::fabs(dot(normal, axis)) < 0.01f) {
            vec3 clipPts[4], clipNrm[4]; vec4 clipPlanes[4];
            vec3 aabbNormal = -normal;
            vec4 refPlane = getAABBReferencePlane(a.a, a.b, aabbNormal);
            ClipPoly poly; poly.numPoints = 2;
            vec3 pa = c.a + normal * c.radius, pb = c.b + normal * c.radius;
            poly.points[0] = VPP{pa, -signedDistanceToPlane(pa, refPlane)};
            poly.points[1] = VPP{pb, -signedDistanceToPlane(pb, refPlane)};
            vec3 aCenter = (a.a + a.b) * 0.5f;
            getAABBClippingPlanes((a.b - a.a) * 0.5f, aabbNormal, clipPts, clipNrm);
            for (int i = 0; i < 4; ++i) { clipPts[i] = clipPts[i] + aCenter; clipPlanes[i] = createPlane(clipPts[i], clipNrm[i]); }
            clipPointsAndBuildContact(poly, clipPlanes, 4, refPlane, out);
        }
    }
    return true;
}

// cylinder vs AABB — 953-1020 (see ora_narrow_cyl below)
bool cylinderCylinder(const World& w, const Shape& a, const Shape& b, ContactManifold& out);
bool cylinderAABB(const World& w, const Shape& c, const Shape& a, ContactManifold& out);

// cylinder vs cylinder — 821-951
bool cylinderCylinder(const World& w, const Shape& a, const Shape& b, ContactManifold& out) {
    vec3 aDir = a.b - a.a;
    vec3 bDir = normalize(b.b - b.a);
    float aDirLength = length(aDir);
    aDir *= 1.f / aDirLength;
    float parallel = dot(aDir, bDir);
    if (std::fabs(parallel) > 0.99f) {
        vec3 pBa = b.a, pBb = b.b;
        if (parallel < 0.f) std::swap(pBa, pBb);
        vec3 referencePoint = a.a;
        float a0 = 0.f, a1 = aDirLength;
        float b0 = dot(aDir, pBa - referencePoint);
        float b1 = dot(aDir, pBb - referencePoint);
        float left = fmax2(a0, b0), right = fmin2(a1, b1);
        if (right < left) return false;
        vec3 contactA0 = referencePoint + left * aDir;
        vec3 contactA1 = referencePoint + right * aDir;
        vec3 contactB0 = closestPoint_PointSegment(contactA0, pBa, pBb);
        vec3 contactB1 = contactB0 + (right - left) * aDir;
        vec3 normal = contactB0 - contactA0;
        float d = length(normal);
        float radiusSum = a.radius + b.radius;
        float penetration = radiusSum - d;
        if (penetration < 0.f) return false;
        float capPenetration = right - left;
        if (capPenetration < penetration) {
            out.numContacts = 1;
            out.depths[0] = capPenetration;
            // NB: the reference subtracts/adds the scalar from every component (vec3 - float via the
            // implicit vec3(float) constructor, collision_narrow.cpp:887,893); restated as written.
            if (b0 > a0) { out.normal = aDir; out.points[0] = a.b - vec3(capPenetration * 0.5f); }
            else { out.normal = -aDir; out.points[0] = a.a + vec3(capPenetration * 0.5f); }
        } else {
            if (d < kEps) { d = 0.f; normal = vec3(0.f, 1.f, 0.f); }
            else normal /= d;
            out.normal = normal;
            out.numContacts = 2;
            out.depths[0] = penetration; out.points[0] = (contactA0 + contactB0) * 0.5f;
            out.depths[1] = penetration; out.points[1] = (contactA1 + contactB1) * 0.5f;
        }
        return true;
    }
    EpaResult epa;
    return gjkEpaSingle(w, a, b, out, epa);
}

// cylinder vs AABB — 953-1020: identical to capsule vs AABB (the cap-contact branch is an empty TODO).
bool cylinderAABB(const World& w, const Shape& c, const Shape& a, ContactManifold& out) { return capsuleAABB(w, c, a, out); }

static Shape toBoxFrame(const Shape& c, const Shape& o) {  // capsule/cylinder endpoints into the OBB's frame (770-775, 1022-1027)
    Shape r = c;
    r.a = conjugate(o.rot) * (c.a - o.a) + o.a;
    r.b = conjugate(o.rot) * (c.b - o.a) + o.a;
    return r;
}
static Shape obbAsAABB(const Shape& o) { Shape r; r.type = T_AABB; r.a = o.a - o.b; r.b = o.a + o.b; return r; }
static void fromBoxFrame(const Shape& o, ContactManifold& out) {
    out.normal = o.rot * out.normal;
    for (uint32_t i = 0; i < out.numContacts; ++i) out.points[i] = o.rot * (out.points[i] - o.a) + o.a;
}

// intersection dispatch — the 21 collision<A,B>() instantiations (collision_narrow.cpp:2473-2570).
// ---------------------------------------------------------------- boolean overlap tests (triggers, force fields)
// overlapCheck — src/physics/collision_narrow.cpp:1586-1689, dispatching to the boolean tests of
// src/physics/bounding_volumes.h:301-363 and bounding_volumes.cpp:704-835, 1079-1244.  A.type <= B.type.
static bool sphereVsSphereB(vec3 ca, float ra, vec3 cb, float rb) {  // bounding_volumes.h:301-307
    vec3 d = ca - cb;
    float dist2 = dot(d, d);
    float radiusSum = ra + rb;
    return dist2 <= radiusSum * radiusSum;
}
static bool sphereVsCylinderB(vec3 sc, float sr, vec3 ca, vec3 cb, float cr) {  // bounding_volumes.cpp:704-724 (compares a squared distance with the radius, as written)
    vec3 ab = cb - ca;
    float t = dot(sc - ca, ab) / squaredLength(ab);
    if (t >= 0.f && t <= 1.f) return sphereVsSphereB(sc, sr, lerp(ca, cb, t), cr);
    vec3 p = (t <= 0.f) ? ca : cb;
    vec3 up = (t <= 0.f) ? -ab : ab;
    vec3 projectedDirToCenter = normalize(cross(cross(up, sc - p), up));
    vec3 endA = p + projectedDirToCenter * cr, endB = p - projectedDirToCenter * cr;
    vec3 closestToSphere = closestPoint_PointSegment(sc, endA, endB);
    float sqDistance = squaredLength(closestToSphere - sc);
    return sqDistance <= sr;
}
static bool sphereVsAABBB(vec3 sc, float sr, vec3 mn, vec3 mx) {  // bounding_volumes.h:320-326
    vec3 p = closestPoint_PointAABB(sc, mn, mx);
    vec3 n = p - sc;
    return squaredLength(n) <= sr * sr;
}
static bool gjkBool(const World& w, const Shape& a, const Shape& b) {
    SupportShape A{&a, a.type == T_HULL ? &w.hulls[a.hull] : nullptr}, B{&b, b.type == T_HULL ? &w.hulls[b.hull] : nullptr};
    GjkSimplex sx;
    return gjkIntersectionTest(A, B, sx);
}
static Shape segmentShapeInBoxFrame(const Shape& c, const Shape& o, Shape& boxOut) {  // capsuleVsOBB / cylinderVsOBB, bounding_volumes.cpp:751-760, 808-817
    boxOut = Shape(); boxOut.type = T_AABB; boxOut.a = o.a - o.b; boxOut.b = o.a + o.b;   // fromCenterRadius
    Shape r = c;
    r.a = conjugate(o.rot) * (c.a - o.a) + o.a;
    r.b = conjugate(o.rot) * (c.b - o.a) + o.a;
    return r;
}
static bool obbVsOBBB(quat arot, vec3 acen, vec3 arad, quat brot, vec3 bcen, vec3 brad) {  // bounding_volumes.cpp:1079-1199
    vec3 ax = arot * vec3(1.f, 0.f, 0.f), ay = arot * vec3(0.f, 1.f, 0.f), az = arot * vec3(0.f, 0.f, 1.f);
    vec3 bx = brot * vec3(1.f, 0.f, 0.f), by = brot * vec3(0.f, 1.f, 0.f), bz = brot * vec3(0.f, 0.f, 1.f);
    mat3 r;
    r.m00 = dot(ax, bx); r.m10 = dot(ay, bx); r.m20 = dot(az, bx);
    r.m01 = dot(ax, by); r.m11 = dot(ay, by); r.m21 = dot(az, by);
    r.m02 = dot(ax, bz); r.m12 = dot(ay, bz); r.m22 = dot(az, bz);
    vec3 tw = bcen - acen;
    vec3 t = conjugate(arot) * tw;
    mat3 q;
    q.m00 = std::fabs(r.m00) + kEps; q.m01 = std::fabs(r.m01) + kEps; q.m02 = std::fabs(r.m02) + kEps;
    q.m10 = std::fabs(r.m10) + kEps; q.m11 = std::fabs(r.m11) + kEps; q.m12 = std::fabs(r.m12) + kEps;
    q.m20 = std::fabs(r.m20) + kEps; q.m21 = std::fabs(r.m21) + kEps; q.m22 = std::fabs(r.m22) + kEps;
    float ra, rb;
    const float ar[3] = {arad.x, arad.y, arad.z}, br[3] = {brad.x, brad.y, brad.z}, tt[3] = {t.x, t.y, t.z};
    for (int i = 0; i < 3; ++i) {
        ra = ar[i]; rb = dot(row(q, i), brad);
        if (ra + rb - std::fabs(tt[i]) < 0.f) return false;
    }
    for (int i = 0; i < 3; ++i) {
        ra = dot(col(q, i), arad); rb = br[i];
        if (ra + rb - std::fabs(dot(col(r, i), t)) < 0.f) return false;
    }
#define ORA_EDGE(RA, RB, D) ra = RA; rb = RB; if (ra + rb - std::fabs(D) < 0.f) return false;
    ORA_EDGE(arad.y * q.m20 + arad.z * q.m10, brad.y * q.m02 + brad.z * q.m01, t.z * r.m10 - t.y * r.m20)
    ORA_EDGE(arad.y * q.m21 + arad.z * q.m11, brad.x * q.m02 + brad.z * q.m00, t.z * r.m11 - t.y * r.m21)
    ORA_EDGE(arad.y * q.m22 + arad.z * q.m12, brad.x * q.m01 + brad.y * q.m00, t.z * r.m12 - t.y * r.m22)
    ORA_EDGE(arad.x * q.m20 + arad.z * q.m00, brad.y * q.m12 + brad.z * q.m11, t.x * r.m20 - t.z * r.m00)
    ORA_EDGE(arad.x * q.m21 + arad.z * q.m01, brad.x * q.m12 + brad.z * q.m10, t.x * r.m21 - t.z * r.m01)
    ORA_EDGE(arad.x * q.m22 + arad.z * q.m02, brad.x * q.m11 + brad.y * q.m10, t.x * r.m22 - t.z * r.m02)
    ORA_EDGE(arad.x * q.m10 + arad.y * q.m00, brad.y * q.m22 + brad.z * q.m21, t.y * r.m00 - t.x * r.m10)
    ORA_EDGE(arad.x * q.m11 + arad.y * q.m01, brad.x * q.m22 + brad.z * q.m20, t.y * r.m01 - t.x * r.m11)
    ORA_EDGE(arad.x * q.m12 + arad.y * q.m02, brad.x * q.m21 + brad.y * q.m20, t.y * r.m02 - t.x * r.m12)
#undef ORA_EDGE
    return true;
}

bool overlapCheck(const World& w, const WorldCollider& A, const WorldCollider& B) {
    const Shape& a = A.s; const Shape& b = B.s;
    switch (a.type) {
        case T_SPHERE:
            switch (b.type) {
                case T_SPHERE: return sphereVsSphereB(a.a, a.radius, b.a, b.radius);
                case T_CAPSULE: return sphereVsSphereB(a.a, a.radius, closestPoint_PointSegment(a.a, b.a, b.b), b.radius);   // bounding_volumes.h:314-318
                case T_CYLINDER: return sphereVsCylinderB(a.a, a.radius, b.a, b.b, b.radius);
                case T_AABB: return sphereVsAABBB(a.a, a.radius, b.a, b.b);
                case T_OBB: return sphereVsAABBB(conjugate(b.rot) * (a.a - b.a) + b.a, a.radius, b.a - b.b, b.a + b.b);   // bounding_volumes.h:328-336
                default: return gjkBool(w, a, b);
            }
        case T_CAPSULE:
            switch (b.type) {
                case T_CAPSULE: { vec3 c1, c2; closestPoint_SegmentSegment(a.a, a.b, b.a, b.b, c1, c2); return sphereVsSphereB(c1, a.radius, c2, b.radius); }
                case T_CYLINDER: { vec3 c1, c2; closestPoint_SegmentSegment(a.a, a.b, b.a, b.b, c1, c2); return sphereVsCylinderB(c1, a.radius, b.a, b.b, b.radius); }
                case T_AABB: return gjkBool(w, a, b);
                case T_OBB: { Shape box; Shape c = segmentShapeInBoxFrame(a, b, box); return gjkBool(w, c, box); }
                default: return gjkBool(w, a, b);
            }
        case T_CYLINDER:
            switch (b.type) {
                case T_CYLINDER: return gjkBool(w, a, b);
                case T_AABB: return gjkBool(w, a, b);
                case T_OBB: { Shape box; Shape c = segmentShapeInBoxFrame(a, b, box); return gjkBool(w, c, box); }
                default: return gjkBool(w, a, b);
            }
        case T_AABB:
            switch (b.type) {
                case T_AABB:   // bounding_volumes.h:352-358
                    if (a.b.x < b.a.x || a.a.x > b.b.x) return false;
                    if (a.b.y < b.a.y || a.a.y > b.b.y) return false;
                    if (a.b.z < b.a.z || a.a.z > b.b.z) return false;
                    return true;
                case T_OBB: return obbVsOBBB(quat(0.f, 0.f, 0.f, 1.f), (a.a + a.b) * 0.5f, (a.b - a.a) * 0.5f, b.rot, b.a, b.b);   // bounding_volumes.h:360-363
                default: return gjkBool(w, a, b);
            }
        case T_OBB:
            if (b.type == T_OBB) return obbVsOBBB(a.rot, a.a, a.b, b.rot, b.a, b.b);
            return gjkBool(w, a, b);
        default:
            return gjkBool(w, a, b);
    }
}

bool intersect(const World& w, const WorldCollider& A, const WorldCollider& B, ContactManifold& out) {
    const Shape& a = A.s; const Shape& b = B.s;
    EpaResult epa;
    switch (a.type) {
        case T_SPHERE:
            switch (b.type) {
                case T_SPHERE: return sphereSphere(a.a, a.radius, b.a, b.radius, out);
                case T_CAPSULE: return sphereSphere(a.a, a.radius, closestPoint_PointSegment(a.a, b.a, b.b), b.radius, out);  // 403-407
                case T_CYLINDER: return sphereCylinder(a.a, a.radius, b.a, b.b, b.radius, out);
                case T_AABB: return sphereAABB(a.a, a.radius, b.a, b.b, out);
                case T_OBB: return sphereOBB(a.a, a.radius, b.rot, b.a, b.b, out);
                case T_HULL: return gjkEpaSingle(w, a, b, out, epa);  // 499-521
            }
            break;
        case T_CAPSULE:
            switch (b.type) {
                case T_CAPSULE: return capsuleVsSegmentShape(a, b, false, out);
                case T_CYLINDER: return capsuleVsSegmentShape(a, b, true, out);
                case T_AABB: return capsuleAABB(w, a, b, out);
                case T_OBB: {  // 770-790
                    Shape c_ = toBoxFrame(a, b); Shape box = obbAsAABB(b);
                    if (capsuleAABB(w, c_, box, out)) { fromBoxFrame(b, out); return true; }
                    return false;
                }
                case T_HULL: return gjkEpaSingle(w, a, b, out, epa);  // 792-818
            }
            break;
        case T_CYLINDER:
            switch (b.type) {
                case T_CYLINDER: return cylinderCylinder(w, a, b, out);
                case T_AABB: return cylinderAABB(w, a, b, out);
                case T_OBB: {  // 1022-1043
                    Shape c_ = toBoxFrame(a, b); Shape box = obbAsAABB(b);
                    if (cylinderAABB(w, c_, box, out)) { fromBoxFrame(b, out); return true; }
                    return false;
                }
                case T_HULL: return gjkEpaSingle(w, a, b, out, epa);  // 1045-1071
            }
            break;
        case T_AABB:
            switch (b.type) {
                case T_AABB: return aabbAABB(a.a, a.b, b.a, b.b, out);
                case T_OBB: return obbOBB(quat(0.f, 0.f, 0.f, 1.f), (a.a + a.b) * 0.5f, (a.b - a.a) * 0.5f, b.rot, b.a, b.b, out);  // 1142-1148
                case T_HULL: return gjkEpaSingle(w, a, b, out, epa);  // 1150-1176
            }
            break;
        case T_OBB:
            switch (b.type) {
                case T_OBB: return obbOBB(a.rot, a.a, a.b, b.rot, b.a, b.b, out);
                case T_HULL: return gjkEpaSingle(w, a, b, out, epa);  // 1529-1555
            }
            break;
        case T_HULL:
            if (b.type == T_HULL) return gjkEpaSingle(w, a, b, out, epa);  // 1558-1584
            break;
    }
    return false;
}


// ---------------------------------------------------------------- ray tests (testPhysicsInteraction)
// ray::intersectSphere / Cylinder / Capsule / AABB / OBB / Hull — src/physics/bounding_volumes.cpp:197-398, 677-702.
// Stated deviation: intersectCylinder leaves outT unset when the origin is inside the infinite cylinder and neither cap is hit
// (the reference then reads it uninitialised); here it is 0 in that case (the ray starts inside the cylinder's slab).
static bool rayPlane(vec3 o, vec3 d, vec3 normal, float pd, float& t) {
    float ndotd = dot(d, normal);
    if (std::fabs(ndotd) < 1e-6f) return false;
    t = -(dot(o, normal) + pd) / ndotd;
    return true;
}
static bool rayDisk(vec3 o, vec3 d, vec3 pos, vec3 normal, float radius, float& t) {
    if (!rayPlane(o, d, normal, -dot(normal, pos), t)) return false;
    return length(o + t * d - pos) <= radius;
}
static bool raySphere(vec3 o, vec3 d, vec3 center, float radius, float& t) {
    vec3 m = o - center;
    float b = dot(m, d), c = dot(m, m) - radius * radius;
    if (c > 0.f && b > 0.f) return false;
    float discr = b * b - c;
    if (discr < 0.f) return false;
    t = -b - std::sqrt(discr);
    if (t < 0.f) t = 0.f;
    return true;
}
static bool rayCylinder(vec3 o, vec3 d, vec3 pa, vec3 pb, float radius, float& t) {
    vec3 axis = pb - pa;
    float height = length(axis);
    quat q = rotateFromTo(axis, vec3(0.f, 1.f, 0.f));
    o = q * (o - pa); d = q * d;
    const float epsilon = 1e-6f;
    float y = -1.f;
    t = 0.f;
    if (o.x * o.x + o.z * o.z > radius * radius) {   // outside the infinite cylinder: the side can be hit
        float a = d.x * d.x + d.z * d.z, b = d.x * o.x + d.z * o.z, c = o.x * o.x + o.z * o.z - radius * radius;
        float delta = b * b - a * c;
        if (delta < epsilon) return false;
        t = (-b - std::sqrt(delta)) / a;
        if (t <= epsilon) return false;
        y = o.y + t * d.y;
    }
    if (y > height + epsilon || y < -epsilon) {       // caps
        float dist;
        if (d.y < 0.f && rayDisk(o, d, vec3(0.f, height, 0.f), vec3(0.f, 1.f, 0.f), radius, dist)) t = dist;
        if (d.y > 0.f && rayDisk(o, d, vec3(0.f, 0.f, 0.f), vec3(0.f, -1.f, 0.f), radius, dist)) t = dist;
        y = o.y + t * d.y;
    }
    return y > -epsilon && y < height + epsilon;
}
static bool rayAABB(vec3 o, vec3 d, vec3 mn, vec3 mx, float& t) {
    vec3 inv(1.f / d.x, 1.f / d.y, 1.f / d.z);
    float tx1 = (mn.x - o.x) * inv.x, tx2 = (mx.x - o.x) * inv.x;
    t = fmin2(tx1, tx2);
    float tmax = fmax2(tx1, tx2);
    float ty1 = (mn.y - o.y) * inv.y, ty2 = (mx.y - o.y) * inv.y;
    t = fmax2(t, fmin2(ty1, ty2)); tmax = fmin2(tmax, fmax2(ty1, ty2));
    float tz1 = (mn.z - o.z) * inv.z, tz2 = (mx.z - o.z) * inv.z;
    t = fmax2(t, fmin2(tz1, tz2)); tmax = fmin2(tmax, fmax2(tz1, tz2));
    return tmax >= t && t > 0.f;
}
static bool pointInTriangle(vec3 point, vec3 a, vec3 b, vec3 c) {   // math.cpp:1273-1290 (sign-bit test)
    vec3 e10 = b - a, e20 = c - a;
    float aa = dot(e10, e10), bb = dot(e10, e20), cc = dot(e20, e20);
    float ac_bb = (aa * cc) - (bb * bb);
    vec3 vp = point - a;
    float dd = dot(vp, e10), ee = dot(vp, e20);
    float x = (dd * cc) - (ee * bb), y = (ee * aa) - (dd * bb), z = x + y - ac_bb;
    uint32_t ux, uy, uz; std::memcpy(&ux, &x, 4); std::memcpy(&uy, &y, 4); std::memcpy(&uz, &z, 4);
    return ((uz & ~(ux | uy)) & 0x80000000u) != 0;
}
static bool rayTriangle(vec3 o, vec3 d, vec3 a, vec3 b, vec3 c, float& t) {
    vec3 normal = noz(cross(b - a, c - a));
    float pd = -dot(normal, a);
    float nDotR = dot(d, normal);
    if (std::fabs(nDotR) <= 1e-6f) return false;
    t = -(dot(o, normal) + pd) / nDotR;
    vec3 q = o + t * d;
    return t >= 0.f && pointInTriangle(q, a, b, c);
}
// One collider in its entity's local frame (collider shapes are stored entity-local; physics.cpp:571-606).
bool rayVsCollider(const World& w, const Shape& s, vec3 o, vec3 d, float& t) {
    switch (s.type) {
        case T_SPHERE: return raySphere(o, d, s.a, s.radius, t);
        case T_CAPSULE: {
            t = FLT_MAX;
            float tt; bool result = false;
            if (rayCylinder(o, d, s.a, s.b, s.radius, tt)) { t = tt; result = true; }
            if (raySphere(o, d, s.a, s.radius, tt)) { t = fmin2(t, tt); result = true; }
            if (raySphere(o, d, s.b, s.radius, tt)) { t = fmin2(t, tt); result = true; }
            return result;
        }
        case T_CYLINDER: return rayCylinder(o, d, s.a, s.b, s.radius, t);
        case T_AABB: return rayAABB(o, d, s.a, s.b, t);
        case T_OBB: return rayAABB(conjugate(s.rot) * (o - s.a), conjugate(s.rot) * d, vec3(0.f) - s.b, vec3(0.f) + s.b, t);
        default: {
            const HullGeometry& g = w.hulls[s.hull];
            vec3 lo = conjugate(s.rot) * (o - s.a), ld = conjugate(s.rot) * d;
            float minT = FLT_MAX; bool result = false;
            for (size_t f = 0; f + 2 < g.tris.size(); f += 3) {
                float tt;
                if (rayTriangle(lo, ld, g.vertices[g.tris[f]], g.vertices[g.tris[f + 1]], g.vertices[g.tris[f + 2]], tt) && tt < minT) { minT = tt; result = true; }
            }
            t = minT;
            return result;
        }
    }
}

}  // namespace ora
