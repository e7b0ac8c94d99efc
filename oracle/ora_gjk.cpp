// ora_gjk.cpp — TEST INFRASTRUCTURE (CPU oracle), see ora_world.h header note.
// GJK (src/physics/collision_gjk.{h,cpp}), EPA (src/physics/collision_epa.{h,cpp}) and the
// GJK/EPA-based cylinder tests (src/physics/collision_narrow.cpp:821-1020).
#include "ora_world.h"
#include <cstring>

namespace ora {

// Support functors — collision_gjk.h:6-100
vec3 support(const SupportShape& sh, vec3 dir) {
    const Shape& s = *sh.s;
    switch (s.type) {
        case T_SPHERE: return normalize(dir) * s.radius + s.a;
        case T_CAPSULE: {
            float distA = dot(dir, s.a), distB = dot(dir, s.b);
            vec3 farther = distA > distB ? s.a : s.b;
            return normalize(dir) * s.radius + farther;
        }
        case T_CYLINDER: {
            float distA = dot(dir, s.a), distB = dot(dir, s.b);
            vec3 farther = distA > distB ? s.a : s.b;
            vec3 n = s.a - s.b;
            vec3 projectedDir = noz(cross(cross(n, dir), n));
            return farther + projectedDir * s.radius;
        }
        case T_AABB:
            return vec3((dir.x < 0.f) ? s.a.x : s.b.x, (dir.y < 0.f) ? s.a.y : s.b.y, (dir.z < 0.f) ? s.a.z : s.b.z);
        case T_OBB: {
            dir = conjugate(s.rot) * dir;
            vec3 r(dir.x < 0.f ? -s.b.x : s.b.x, dir.y < 0.f ? -s.b.y : s.b.y, dir.z < 0.f ? -s.b.z : s.b.z);
            return s.a + s.rot * r;
        }
        case T_HULL: {
            dir = conjugate(s.rot) * dir;
            vec3 result;
            float maxDist = -FLT_MAX;
            for (const vec3& v : sh.g->vertices) {
                float d = dot(dir, v);
                if (d > maxDist) { maxDist = d; result = v; }
            }
            return s.a + s.rot * result;
        }
    }
    return vec3(0.f);
}

static GjkSimplexPoint supportPoint(const SupportShape& A, const SupportShape& B, vec3 dir) {  // collision_gjk.h:163-169
    GjkSimplexPoint p;
    p.shapeAPoint = support(A, dir);
    p.shapeBPoint = support(B, -dir);
    p.minkowski = p.shapeAPoint - p.shapeBPoint;
    return p;
}
static inline vec3 crossABA(vec3 a, vec3 b) { return cross(cross(a, b), a); }

enum { GJK_STOP, GJK_DONT_STOP, GJK_ERROR };

// updateGJKSimplex — collision_gjk.cpp:6-212 (goto structure kept as labelled fallthrough)
static int updateGJKSimplex(GjkSimplex& s, const GjkSimplexPoint& a, vec3& dir) {
    if (s.numPoints == 2) {
        vec3 ao = -a.minkowski;
        vec3 ab = s.b.minkowski - a.minkowski;
        vec3 ac = s.c.minkowski - a.minkowski;
        vec3 abc = cross(ab, ac);
        vec3 abp = cross(ab, abc);
        if (dot(ao, abp) > 0.f) { s.c = a; dir = crossABA(ab, ao); return GJK_DONT_STOP; }
        vec3 acp = cross(abc, ac);
        if (dot(ao, acp) > 0.f) { s.b = a; dir = crossABA(ac, ao); return GJK_DONT_STOP; }
        if (dot(ao, abc) >= 0.f) { s.d = s.b; s.b = a; s.numPoints = 3; dir = abc; return GJK_DONT_STOP; }
        if (dot(ao, -abc) >= 0.f) { s.d = s.c; s.c = s.b; s.b = a; s.numPoints = 3; dir = -abc; return GJK_DONT_STOP; }
        return GJK_ERROR;
    }
    if (s.numPoints == 3) {
        vec3 ao = -a.minkowski;
        vec3 ab = s.b.minkowski - a.minkowski;
        vec3 ac = s.c.minkowski - a.minkowski;
        vec3 ad = s.d.minkowski - a.minkowski;
        vec3 bcd = cross(s.c.minkowski - s.b.minkowski, s.d.minkowski - s.b.minkowski);
        if (dot(bcd, dir) > 0.00001f || dot(bcd, s.b.minkowski) < -0.00001f) return GJK_ERROR;
        vec3 abc = cross(ac, ab), abd = cross(ab, ad), adc = cross(ad, ac);
        const int overABC = 1, overABD = 2, overADC = 4;
        int flags = 0;
        flags |= (dot(abc, ao) > 0.f) ? overABC : 0;
        flags |= (dot(abd, ao) > 0.f) ? overABD : 0;
        flags |= (dot(adc, ao) > 0.f) ? overADC : 0;
        if (flags == (overABC | overABD | overADC)) return GJK_ERROR;
        if (flags == 0) return GJK_STOP;
        // entry label: 1 = overABC1, 2 = overABC2, 3 = overABD1, 4 = overABD2, 5 = overADC1, 6 = overADC2
        int label = 0;
        if (flags == overABC) label = 1;
        else if (flags == overABD) label = 3;
        else if (flags == overADC) label = 5;
        else if (flags == (overABC | overABD)) label = (dot(cross(abc, ab), ao) > 0.f) ? 3 : 2;
        else if (flags == (overABD | overADC)) label = (dot(cross(abd, ad), ao) > 0.f) ? 5 : 4;
        else if (flags == (overADC | overABC)) label = (dot(cross(adc, ac), ao) > 0.f) ? 1 : 6;
        switch (label) {
            case 1:
                if (dot(cross(abc, ab), ao) > 0.f) { s.c = a; s.numPoints = 2; dir = crossABA(ab, ao); return GJK_DONT_STOP; }
                // fallthrough
            case 2:
                if (dot(cross(ac, abc), ao) > 0.f) { s.b = a; s.numPoints = 2; dir = crossABA(ac, ao); return GJK_DONT_STOP; }
                s.d = a; dir = abc; return GJK_DONT_STOP;
            case 3:
                if (dot(cross(abd, ad), ao) > 0.f) { s.b = s.d; s.c = a; s.numPoints = 2; dir = crossABA(ad, ao); return GJK_DONT_STOP; }
                // fallthrough
            case 4:
                if (dot(cross(ab, abd), ao) > 0.f) { s.c = a; s.numPoints = 2; dir = crossABA(ab, ao); return GJK_DONT_STOP; }
                s.c = a; dir = abd; return GJK_DONT_STOP;
            case 5:
                if (dot(cross(adc, ac), ao) > 0.f) { s.b = a; s.numPoints = 2; dir = crossABA(ac, ao); return GJK_DONT_STOP; }
                // fallthrough
            case 6:
                if (dot(cross(ad, adc), ao) > 0.f) { s.b = a; s.c = s.d; s.numPoints = 2; dir = crossABA(ad, ao); return GJK_DONT_STOP; }
                s.b = a; dir = adc; return GJK_DONT_STOP;
        }
        return GJK_ERROR;
    }
    return GJK_ERROR;
}

// gjkIntersectionTest — collision_gjk.h:182-238.  The reference loop is unbounded; a 64-iteration
// guard (treated as "no collision", like its unexpected-error path) keeps a GPU thread bounded.
bool gjkIntersectionTest(const SupportShape& A, const SupportShape& B, GjkSimplex& sx) {
    vec3 dir(1.f, 0.1f, -0.2f);
    sx.numPoints = 0;
    sx.c = supportPoint(A, B, dir);
    if (dot(sx.c.minkowski, dir) < 0.f) return false;
    dir = -sx.c.minkowski;
    sx.b = supportPoint(A, B, dir);
    if (dot(sx.b.minkowski, dir) < 0.f) return false;
    dir = crossABA(sx.c.minkowski - sx.b.minkowski, -sx.b.minkowski);
    sx.numPoints = 2;
    for (int guard = 0; guard < 64; ++guard) {
        if (squaredLength(dir) < 0.0001f) return false;
        GjkSimplexPoint a = supportPoint(A, B, dir);
        if (dot(a.minkowski, dir) < 0.f) return false;
        int r = updateGJKSimplex(sx, a, dir);
        if (r == GJK_STOP) { sx.a = a; sx.numPoints = 4; return true; }
        if (r == GJK_ERROR) return false;
    }
    return false;
}

// ---------------------------------------------------------------- EPA

// Capacities: the reference uses 1024/1024/1024 + 128 border edges (collision_epa.h:48-50,
// collision_epa.cpp:141).  With <= 20 iterations at most 24 points exist and each iteration adds
// (horizon size <= current point count) triangles and edges, so 4 + sum(4..23) = 274 triangles and
// 276 edges bound the arrays: the capacities below can never trigger "out of memory" differently
// from the reference's.
enum { EPA_MAX_POINTS = 24, EPA_MAX_TRIS = 288, EPA_MAX_EDGES = 288, EPA_MAX_BORDER = 32 };
struct EpaTri { uint16_t a, b, c, eA, eB, eC; vec3 normal; float dist; };
struct EpaEdge { uint16_t a, b, tA, tB; };
struct EpaSimplex {
    GjkSimplexPoint points[EPA_MAX_POINTS];
    EpaTri tris[EPA_MAX_TRIS];
    EpaEdge edges[EPA_MAX_EDGES];
    bool active[EPA_MAX_TRIS];
    uint16_t numTris = 0, numPoints = 0, numEdges = 0;
};
struct TriInfo { vec3 normal; float dist; };

static TriInfo getTriangleInfo(const GjkSimplexPoint& a, const GjkSimplexPoint& b, const GjkSimplexPoint& c) {  // collision_epa.cpp:5-11
    TriInfo r;
    r.normal = normalize(cross(b.minkowski - a.minkowski, c.minkowski - a.minkowski));
    r.dist = dot(r.normal, a.minkowski);
    return r;
}
static uint16_t pushPoint(EpaSimplex& s, const GjkSimplexPoint& p) {
    if (s.numPoints >= EPA_MAX_POINTS) return UINT16_MAX;
    s.points[s.numPoints] = p; return s.numPoints++;
}
static uint16_t pushTriangle(EpaSimplex& s, uint16_t a, uint16_t b, uint16_t c, uint16_t eA, uint16_t eB, uint16_t eC, TriInfo info) {
    if (s.numTris >= EPA_MAX_TRIS) return UINT16_MAX;
    uint16_t i = s.numTris++;
    s.active[i] = true;
    EpaTri& t = s.tris[i];
    t.a = a; t.b = b; t.c = c; t.eA = eA; t.eB = eB; t.eC = eC; t.normal = info.normal; t.dist = info.dist;
    return i;
}
static uint16_t pushEdge(EpaSimplex& s, uint16_t a, uint16_t b, uint16_t tA, uint16_t tB) {
    if (s.numEdges >= EPA_MAX_EDGES) return UINT16_MAX;
    uint16_t i = s.numEdges++;
    s.edges[i] = EpaEdge{a, b, tA, tB};
    return i;
}
static uint32_t findTriangleClosestToOrigin(const EpaSimplex& s) {  // collision_epa.cpp:95-115
    uint32_t closest = 0xFFFFFFFFu; float minDistance = FLT_MAX;
    for (uint32_t i = 0; i < s.numTris; ++i)
        if (s.active[i] && s.tris[i].dist < minDistance) { minDistance = s.tris[i].dist; closest = i; }
    return closest;
}
// addNewPointAndUpdate — collision_epa.cpp:117-240
// Where the reference's code reads what it never wrote — on a degenerate polytope (zero-area faces have NaN normals and are never "seen", so the faces a new point sees need
// not form one loop) a new face's third edge is looked up in newEdgePerPoint[] for a point no horizon edge started from, and an edge's first face may stay unset — the
// reference's behaviour is undefined (it indexes its 1024-entry arrays with whatever the stack held).  Oracle and product define it the same way instead: such an index is
// 0xFFFF, "no edge" / "no face": it is never dereferenced, counts no reference, and a missing face counts as inactive.
static bool addNewPointAndUpdate(EpaSimplex& s, const GjkSimplexPoint& np) {
    uint8_t edgeRefs[EPA_MAX_EDGES];
    std::memset(edgeRefs, 0, sizeof(edgeRefs));
    auto ref = [&](uint16_t e) { if (e < EPA_MAX_EDGES) ++edgeRefs[e]; };
    auto isActive = [&](uint16_t t) { return t < EPA_MAX_TRIS && s.active[t]; };
    for (uint32_t i = 0; i < s.numTris; ++i) {
        if (!s.active[i]) continue;
        EpaTri& t = s.tris[i];
        float d = dot(t.normal, np.minkowski - s.points[t.a].minkowski);
        if (d > 0.f) { ref(t.eA); ref(t.eB); ref(t.eC); s.active[i] = false; }
    }
    uint16_t border[EPA_MAX_BORDER]; uint32_t numBorder = 0;
    for (uint32_t i = 0; i < s.numEdges; ++i)
        if (edgeRefs[i] == 1) { if (numBorder >= EPA_MAX_BORDER) return false; border[numBorder++] = (uint16_t)i; }
    uint16_t newEdgePerPoint[EPA_MAX_POINTS];
    for (uint16_t& e : newEdgePerPoint) e = UINT16_MAX;
    uint16_t npi = pushPoint(s, np);
    if (npi == UINT16_MAX) return false;
    uint16_t triOffset = s.numTris;
    for (uint32_t i = 0; i < numBorder; ++i) {
        uint16_t ei = border[i];
        EpaEdge& e = s.edges[ei];
        bool triAActive = isActive(e.tA), triBActive = isActive(e.tB);
        uint16_t pointToConnect = triBActive ? e.a : e.b;
        uint16_t triIndex = s.numTris;
        uint16_t newEdge = pushEdge(s, pointToConnect, npi, UINT16_MAX, s.numTris);
        if (newEdge == UINT16_MAX) return false;
        newEdgePerPoint[pointToConnect] = newEdge;
        uint16_t bI = pointToConnect, cI = triBActive ? e.b : e.a;
        uint16_t test = pushTriangle(s, npi, bI, cI, ei, UINT16_MAX, newEdge, getTriangleInfo(np, s.points[bI], s.points[cI]));
        if (test == UINT16_MAX) return false;
        if (triAActive) s.edges[ei].tB = triIndex; else s.edges[ei].tA = triIndex;
    }
    for (uint32_t i = 0; i < numBorder; ++i) {
        EpaEdge& e = s.edges[border[i]];
        bool triBNew = e.tB >= triOffset;
        uint16_t pointToConnect = triBNew ? e.a : e.b;
        uint16_t other = newEdgePerPoint[pointToConnect];
        uint16_t triIndex = (uint16_t)(i + triOffset);
        s.tris[triIndex].eB = other;
        if (other < EPA_MAX_EDGES) s.edges[other].tA = triIndex;
    }
    return true;
}

static vec3 getBarycentricCoordinates(vec3 a, vec3 b, vec3 c, vec3 p) {  // src/core/math.cpp:1391-1407
    vec3 v0 = b - a, v1 = c - a, v2 = p - a;
    float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    denom = (std::fabs(denom) < kEps) ? 1.f : denom;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    return vec3(u, v, w);
}

// epaCollisionInfo — collision_epa.h:96-168.  Returns 1 success, 2 out of memory, 3 max iterations.
int epaCollisionInfo(const GjkSimplex& g, const SupportShape& A, const SupportShape& B, EpaResult& out) {
    EpaSimplex s;
    pushPoint(s, g.a); pushPoint(s, g.b); pushPoint(s, g.c); pushPoint(s, g.d);
    pushTriangle(s, 0, 1, 3, 4, 3, 0, getTriangleInfo(g.a, g.b, g.d));
    pushTriangle(s, 1, 2, 3, 5, 4, 1, getTriangleInfo(g.b, g.c, g.d));
    pushTriangle(s, 2, 0, 3, 3, 5, 2, getTriangleInfo(g.c, g.a, g.d));
    pushTriangle(s, 0, 2, 1, 1, 0, 2, getTriangleInfo(g.a, g.c, g.b));
    pushEdge(s, 0, 1, 0, 3); pushEdge(s, 1, 2, 1, 3); pushEdge(s, 2, 0, 2, 3);
    pushEdge(s, 0, 3, 2, 0); pushEdge(s, 1, 3, 0, 1); pushEdge(s, 2, 3, 1, 2);
    uint32_t closest = 0;
    int rc = 3;
    for (uint32_t it = 0; it < 20; ++it) {
        uint32_t prev = closest;
        closest = findTriangleClosestToOrigin(s);
        if (closest == 0xFFFFFFFFu) { closest = prev; break; }  // degenerate polytope: keep the last face (the reference asserts here)
        EpaTri& t = s.tris[closest];
        GjkSimplexPoint a = supportPoint(A, B, t.normal);
        float d = dot(a.minkowski, t.normal);
        if (d - t.dist < 0.01f) { rc = 1; break; }
        if (!addNewPointAndUpdate(s, a)) { rc = 2; break; }
    }
    EpaTri& t = s.tris[closest];
    GjkSimplexPoint& a = s.points[t.a]; GjkSimplexPoint& b = s.points[t.b]; GjkSimplexPoint& c = s.points[t.c];
    vec3 bc = getBarycentricCoordinates(a.minkowski, b.minkowski, c.minkowski, t.normal * t.dist);
    vec3 pointA = bc.x * a.shapeAPoint + bc.y * b.shapeAPoint + bc.z * c.shapeAPoint;
    vec3 pointB = bc.x * a.shapeBPoint + bc.y * b.shapeBPoint + bc.z * c.shapeBPoint;
    out.point = 0.5f * (pointA + pointB);
    out.normal = t.normal;
    out.penetrationDepth = t.dist;
    return rc;
}

}  // namespace ora
