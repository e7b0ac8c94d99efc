// ora_joints.cpp — TEST INFRASTRUCTURE (CPU oracle), see ora_world.h header note.
// Joint constraints: storage, the add*ConstraintFromGlobalPoints helpers (src/physics/physics.cpp:128-333)
// and the scalar initialize/solve routines of src/physics/constraints.cpp (distance 189-264, ball 460-528,
// fixed 736-823, hinge 1079-1307, cone-twist 1782-2070, slider 2638-2846), solved per type in the order
// distance -> ball -> fixed -> hinge -> cone-twist -> slider (constraints.cpp:3764-3769).
#include "ora_world.h"
#include <algorithm>
#include <type_traits>
#include <cstring>

namespace ora {

static const float BETA_DISTANCE = 0.1f, BETA_BALL = 0.1f, BETA_SLIDER = 0.1f, BETA_HINGE_ROT = 0.3f, BETA_HINGE_LIMIT = 0.1f,
                   BETA_TWIST_LIMIT = 0.1f, BETA_SLIDER_LIMIT = 0.1f, DT_THRESHOLD = 1e-5f;  // constraints.cpp:9-17

struct vec2 { float x, y; };
struct mat2 { float m00, m01, m10, m11; };
static inline vec2 solve2(const mat2& A, vec2 b) { vec2 r; solveLinearSystem2(A.m00, A.m01, A.m10, A.m11, b.x, b.y, r.x, r.y); return r; }

struct DistanceUpd { uint32_t a, b; vec3 rA, rB, iwA, iwB, u; float bias, effMass; };
struct BallUpd { uint32_t a, b; vec3 rA, rB, bias; mat3 invEffMass; };
struct FixedUpd { uint32_t a, b; vec3 rA, rB, tBias, rBias; mat3 invEffT, invEffR; };
struct HingeUpd {
    uint32_t a, b; vec3 rA, rB, tBias; mat3 invEffT; vec3 bxa, cxa; mat2 invEffR; vec2 rBias;
    bool solveLimit, solveMotor; vec3 axis; float limitImpulse, effAxial, limitSign, maxMotorImpulse, motorImpulse, motorVelocity, limitBias;
    vec3 mlA, mlB;
};
struct ConeUpd {
    uint32_t a, b; vec3 rA, rB, bias; mat3 invEff;
    bool solveSwingLimit, solveSwingMotor, solveTwistLimit, solveTwistMotor;
    float swingImpulse; vec3 swingAxis; float effSwingLimit, swingLimitBias; vec3 slA, slB;
    float maxSwingMotorImpulse, swingMotorImpulse, swingMotorVelocity, effSwingMotor; vec3 swingMotorAxis, smA, smB;
    float twistImpulse; vec3 twistAxis; float effTwist, twistLimitSign, maxTwistMotorImpulse, twistMotorImpulse, twistMotorVelocity, twistLimitBias;
    vec3 tmA, tmB;
};
struct SliderUpd {
    uint32_t a, b; vec3 tangent, bitangent, rBxt, rBxb, rAuxt, rAuxb; mat2 invEffT; mat3 invEffR; vec2 tBias; vec3 rBias;
    vec3 axis; bool solveLimit, solveMotor; float limitImpulse; vec3 rAuxs, rBxs; float effAxial, limitSign, limitBias; vec3 llA, llB;
    float maxMotorImpulse, motorImpulse, motorVelocity;
};

// Dense arrays in EnTT pool order (append on create, swap-and-pop on destroy); the public handle of a constraint stays valid
// across deletions through handle -> dense index.  `ents` / `seq`: the two entities and a world-wide creation counter
// (deleteAllConstraintsFromEntity walks an entity's edge list, newest constraint first: physics.cpp:87-111, 523-539).
template <typename P> struct JointList {
    std::vector<P> pods; std::vector<Pair> bodies; std::vector<uint32_t> order;
    std::vector<uint32_t> handleAt; std::vector<int32_t> denseOf; std::vector<Pair> ents; std::vector<uint64_t> seq;
    int dense(uint32_t handle) const { return handle < denseOf.size() ? denseOf[handle] : -1; }
    bool destroy(uint32_t handle) {   // registry.destroy(constraintEntity): swap-and-pop in the component pool
        int d = dense(handle);
        if (d < 0) return false;
        size_t last = pods.size() - 1;
        pods[d] = pods[last]; bodies[d] = bodies[last]; handleAt[d] = handleAt[last]; ents[d] = ents[last]; seq[d] = seq[last];
        denseOf[handleAt[d]] = d;
        pods.pop_back(); bodies.pop_back(); handleAt.pop_back(); ents.pop_back(); seq.pop_back();
        denseOf[handle] = -1;
        return true;
    }
    void clear() { pods.clear(); bodies.clear(); handleAt.clear(); ents.clear(); seq.clear(); std::fill(denseOf.begin(), denseOf.end(), -1); }
};

struct JointStore {
    JointList<mi_distance_constraint> distance; JointList<mi_ball_constraint> ball; JointList<mi_fixed_constraint> fixed;
    JointList<mi_hinge_constraint> hinge; JointList<mi_cone_twist_constraint> cone; JointList<mi_slider_constraint> slider;
    std::vector<DistanceUpd> uDistance; std::vector<BallUpd> uBall; std::vector<FixedUpd> uFixed;
    std::vector<HingeUpd> uHinge; std::vector<ConeUpd> uCone; std::vector<SliderUpd> uSlider;
    bool orderDirty = true;
    uint64_t nextSeq = 0;
};

JointStore* jointsCreate() { return new JointStore(); }
void jointsDestroy(JointStore* j) { delete j; }
uint32_t jointsCount(const World& w) {
    const JointStore& j = *w.joints;
    return (uint32_t)(j.distance.pods.size() + j.ball.pods.size() + j.fixed.pods.size() + j.hinge.pods.size() + j.cone.pods.size() + j.slider.pods.size());
}

void jointsIslandRoots(const World& w, std::vector<uint32_t>& root) {
    const uint32_t nb = (uint32_t)w.bodies.size();
    std::vector<uint32_t> parent(nb);
    for (uint32_t i = 0; i < nb; ++i) parent[i] = i;
    auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    const JointStore& j = *w.joints;
    auto link = [&](const auto& l) { for (const Pair& b : l.bodies) { uint32_t x = find(b.a), y = find(b.b); if (x != y) parent[std::max(x, y)] = std::min(x, y); } };   // the root is the lowest index
    link(j.distance); link(j.ball); link(j.fixed); link(j.hinge); link(j.cone); link(j.slider);
    root.resize(nb);
    for (uint32_t i = 0; i < nb; ++i) root[i] = find(i);
}
int jointsLoadPods(World& w, const uint8_t*& p, const uint8_t* end, const uint32_t counts[6]) {
    JointStore& j = *w.joints;
    auto load = [&](auto& list, uint32_t n) {
        if (n != list.pods.size()) return false;
        const size_t bytes = list.pods.size() * sizeof(list.pods[0]);
        if ((size_t)(end - p) < bytes) return false;
        if (bytes) std::memcpy(list.pods.data(), p, bytes);
        p += bytes;
        return true;
    };
    bool okay = load(j.distance, counts[0]) && load(j.ball, counts[1]) && load(j.fixed, counts[2]) && load(j.hinge, counts[3]) && load(j.cone, counts[4]) && load(j.slider, counts[5]);
    return okay ? MI_OK : MI_ERR_INVALID_ARGUMENT;
}

void jointsSavePods(const World& w, std::vector<uint8_t>& out, uint32_t counts[6]) {
    const JointStore& j = *w.joints;
    auto save = [&](const auto& list, uint32_t& n) {
        n = (uint32_t)list.pods.size();
        if (n) { const uint8_t* b = reinterpret_cast<const uint8_t*>(list.pods.data()); out.insert(out.end(), b, b + list.pods.size() * sizeof(list.pods[0])); }
    };
    save(j.distance, counts[0]); save(j.ball, counts[1]); save(j.fixed, counts[2]); save(j.hinge, counts[3]); save(j.cone, counts[4]); save(j.slider, counts[5]);
}

static inline vec3 v3(const float* f) { return vec3(f[0], f[1], f[2]); }
static inline quat q4(const float* f) { return quat(f[0], f[1], f[2], f[3]); }
static inline void st3(float* f, vec3 v) { f[0] = v.x; f[1] = v.y; f[2] = v.z; }
static inline void st4(float* f, quat q) { f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w; }

// packed size, or the reference's own sizeof (a leading quat pads fixed / slider by 8 bytes: include/mi_constraints.h "Reference layout")
template <typename P> static bool podSizeOk(uint32_t bytes) {
    const uint32_t ref = (std::is_same<P, mi_fixed_constraint>::value || std::is_same<P, mi_slider_constraint>::value) ? (uint32_t)((sizeof(P) + 15u) & ~15u) : (uint32_t)sizeof(P);
    return bytes == sizeof(P) || bytes == ref;
}
template <typename P>
static int addTo(World& w, JointList<P>& l, uint32_t ea, uint32_t eb, const void* pod, uint32_t bytes, uint32_t* out) {
    if (!podSizeOk<P>(bytes)) return MI_ERR_INVALID_ARGUMENT;
    if (ea >= w.entities.size() || eb >= w.entities.size() || w.entities[ea].rb < 0 || w.entities[eb].rb < 0) return MI_ERR_INVALID_ARGUMENT;
    P p; std::memcpy(&p, pod, sizeof(P));
    const uint32_t handle = (uint32_t)l.denseOf.size();
    if (out) *out = handle;
    l.denseOf.push_back((int32_t)l.pods.size()); l.handleAt.push_back(handle);
    l.ents.push_back(Pair{ea, eb}); l.seq.push_back(w.joints->nextSeq++);
    l.pods.push_back(p);
    l.bodies.push_back(Pair{(uint32_t)w.entities[ea].rb, (uint32_t)w.entities[eb].rb});
    w.joints->orderDirty = true;
    return MI_OK;
}
int jointsAdd(World& w, uint32_t type, uint32_t ea, uint32_t eb, const void* pod, uint32_t bytes, uint32_t* out) {
    JointStore& j = *w.joints;
    switch (type) {
        case MI_CONSTRAINT_DISTANCE: return addTo(w, j.distance, ea, eb, pod, bytes, out);
        case MI_CONSTRAINT_BALL: return addTo(w, j.ball, ea, eb, pod, bytes, out);
        case MI_CONSTRAINT_FIXED: return addTo(w, j.fixed, ea, eb, pod, bytes, out);
        case MI_CONSTRAINT_HINGE: return addTo(w, j.hinge, ea, eb, pod, bytes, out);
        case MI_CONSTRAINT_CONE_TWIST: return addTo(w, j.cone, ea, eb, pod, bytes, out);
        case MI_CONSTRAINT_SLIDER: return addTo(w, j.slider, ea, eb, pod, bytes, out);
    }
    return MI_ERR_INVALID_ARGUMENT;
}
template <typename P> static int updIn(JointList<P>& l, uint32_t id, const void* pod, uint32_t bytes) {
    if (!podSizeOk<P>(bytes) || l.dense(id) < 0) return MI_ERR_INVALID_ARGUMENT;
    std::memcpy(&l.pods[l.dense(id)], pod, sizeof(P)); return MI_OK;
}
template <typename P> static int getIn(JointList<P>& l, uint32_t id, void* pod, uint32_t bytes) {
    if (!podSizeOk<P>(bytes) || l.dense(id) < 0) return MI_ERR_INVALID_ARGUMENT;
    std::memcpy(pod, &l.pods[l.dense(id)], sizeof(P)); return MI_OK;
}
// deleteConstraint / deleteAllConstraints / deleteAllConstraintsFromEntity — src/physics/physics.cpp:443-539
int jointsDestroy(World& w, uint32_t type, uint32_t id) {
    JointStore& j = *w.joints;
    bool ok = false;
    switch (type) {
        case MI_CONSTRAINT_DISTANCE: ok = j.distance.destroy(id); break;
        case MI_CONSTRAINT_BALL: ok = j.ball.destroy(id); break;
        case MI_CONSTRAINT_FIXED: ok = j.fixed.destroy(id); break;
        case MI_CONSTRAINT_HINGE: ok = j.hinge.destroy(id); break;
        case MI_CONSTRAINT_CONE_TWIST: ok = j.cone.destroy(id); break;
        case MI_CONSTRAINT_SLIDER: ok = j.slider.destroy(id); break;
    }
    j.orderDirty = true;
    return ok ? MI_OK : MI_ERR_INVALID_ARGUMENT;
}
void jointsDestroyAll(World& w) {
    JointStore& j = *w.joints;
    j.distance.clear(); j.ball.clear(); j.fixed.clear(); j.hinge.clear(); j.cone.clear(); j.slider.clear();
    j.orderDirty = true;
}
int jointsDestroyOfEntity(World& w, uint32_t entity) {
    JointStore& j = *w.joints;
    struct Hit { uint64_t seq; uint32_t type, handle; };
    std::vector<Hit> hits;
    auto scan = [&](uint32_t type, const auto& l) { for (size_t d = 0; d < l.pods.size(); ++d) if (l.ents[d].a == entity || l.ents[d].b == entity) hits.push_back(Hit{l.seq[d], type, l.handleAt[d]}); };
    scan(MI_CONSTRAINT_DISTANCE, j.distance); scan(MI_CONSTRAINT_BALL, j.ball); scan(MI_CONSTRAINT_FIXED, j.fixed);
    scan(MI_CONSTRAINT_HINGE, j.hinge); scan(MI_CONSTRAINT_CONE_TWIST, j.cone); scan(MI_CONSTRAINT_SLIDER, j.slider);
    std::sort(hits.begin(), hits.end(), [](const Hit& x, const Hit& y) { return x.seq > y.seq; });   // the edge list is newest first
    for (const Hit& h : hits) jointsDestroy(w, h.type, h.handle);
    return MI_OK;
}
// A rigid body moved inside the body pool (swap-and-pop on deletion): the reference derives a constraint's body pair from its
// entities every step (getConstraintBodyPairs, physics.cpp:789-806), here the cached pair is re-pointed.
void jointsRemapBody(World& w, uint32_t from, uint32_t to) {
    JointStore& j = *w.joints;
    auto fix = [&](auto& l) { for (Pair& b : l.bodies) { if (b.a == from) b.a = to; if (b.b == from) b.b = to; } };
    fix(j.distance); fix(j.ball); fix(j.fixed); fix(j.hinge); fix(j.cone); fix(j.slider);
    j.orderDirty = true;
}
int jointsUpdate(World& w, uint32_t type, uint32_t id, const void* pod, uint32_t bytes) {
    JointStore& j = *w.joints;
    switch (type) {
        case MI_CONSTRAINT_DISTANCE: return updIn(j.distance, id, pod, bytes);
        case MI_CONSTRAINT_BALL: return updIn(j.ball, id, pod, bytes);
        case MI_CONSTRAINT_FIXED: return updIn(j.fixed, id, pod, bytes);
        case MI_CONSTRAINT_HINGE: return updIn(j.hinge, id, pod, bytes);
        case MI_CONSTRAINT_CONE_TWIST: return updIn(j.cone, id, pod, bytes);
        case MI_CONSTRAINT_SLIDER: return updIn(j.slider, id, pod, bytes);
    }
    return MI_ERR_INVALID_ARGUMENT;
}
int jointsGet(World& w, uint32_t type, uint32_t id, void* pod, uint32_t bytes) {
    JointStore& j = *w.joints;
    switch (type) {
        case MI_CONSTRAINT_DISTANCE: return getIn(j.distance, id, pod, bytes);
        case MI_CONSTRAINT_BALL: return getIn(j.ball, id, pod, bytes);
        case MI_CONSTRAINT_FIXED: return getIn(j.fixed, id, pod, bytes);
        case MI_CONSTRAINT_HINGE: return getIn(j.hinge, id, pod, bytes);
        case MI_CONSTRAINT_CONE_TWIST: return getIn(j.cone, id, pod, bytes);
        case MI_CONSTRAINT_SLIDER: return getIn(j.slider, id, pod, bytes);
    }
    return MI_ERR_INVALID_ARGUMENT;
}

// add*ConstraintFromGlobalPoints — src/physics/physics.cpp:147-333.  Uses the entities' transform_component.
int jointsAddFromGlobal(World& w, uint32_t type, uint32_t ea, uint32_t eb, const float* anchor, const float* axisIn, float l0, float l1, uint32_t* out) {
    if (ea >= w.entities.size() || eb >= w.entities.size()) return MI_ERR_INVALID_ARGUMENT;
    const Entity& A = w.entities[ea]; const Entity& B = w.entities[eb];
    auto invPos = [](const Entity& e, vec3 p) { vec3 r = conjugate(e.rotation) * (p - e.position); return vec3(r.x / 1.f, r.y / 1.f, r.z / 1.f); };
    auto invDir = [](const Entity& e, vec3 d) { return conjugate(e.rotation) * d; };
    vec3 ga = v3(anchor);
    vec3 gx = axisIn ? v3(axisIn) : vec3(0.f);
    switch (type) {
        case MI_CONSTRAINT_DISTANCE: {  // anchor = globalAnchorA, axis = globalAnchorB
            mi_distance_constraint c;
            st3(c.local_anchor_a, invPos(A, ga)); st3(c.local_anchor_b, invPos(B, gx));
            c.global_length = length(ga - gx);
            return jointsAdd(w, type, ea, eb, &c, sizeof(c), out);
        }
        case MI_CONSTRAINT_BALL: {
            mi_ball_constraint c; st3(c.local_anchor_a, invPos(A, ga)); st3(c.local_anchor_b, invPos(B, ga));
            return jointsAdd(w, type, ea, eb, &c, sizeof(c), out);
        }
        case MI_CONSTRAINT_FIXED: {
            mi_fixed_constraint c; st3(c.local_anchor_a, invPos(A, ga)); st3(c.local_anchor_b, invPos(B, ga));
            st4(c.initial_inv_rotation_difference, conjugate(B.rotation) * A.rotation);
            return jointsAdd(w, type, ea, eb, &c, sizeof(c), out);
        }
        case MI_CONSTRAINT_HINGE: {
            mi_hinge_constraint c;
            st3(c.local_anchor_a, invPos(A, ga)); st3(c.local_anchor_b, invPos(B, ga));
            vec3 axA = invDir(A, gx), axB = invDir(B, gx);
            st3(c.local_hinge_axis_a, axA); st3(c.local_hinge_axis_b, axB);
            vec3 t, bt; getTangents(axA, t, bt);
            st3(c.local_hinge_tangent_a, t); st3(c.local_hinge_bitangent_a, bt);
            st3(c.local_hinge_tangent_b, conjugate(B.rotation) * (A.rotation * t));
            c.min_rotation_limit = l0; c.max_rotation_limit = l1;
            c.motor_type = MI_MOTOR_VELOCITY; c.motor_velocity_or_target_angle = 0.f; c.max_motor_torque = -1.f;
            return jointsAdd(w, type, ea, eb, &c, sizeof(c), out);
        }
        case MI_CONSTRAINT_CONE_TWIST: {
            mi_cone_twist_constraint c;
            st3(c.local_anchor_a, invPos(A, ga)); st3(c.local_anchor_b, invPos(B, ga));
            c.swing_limit = l0; c.twist_limit = l1;
            vec3 axA = invDir(A, gx), axB = invDir(B, gx);
            st3(c.local_limit_axis_a, axA); st3(c.local_limit_axis_b, axB);
            vec3 t, bt; getTangents(axA, t, bt);
            st3(c.local_limit_tangent_a, t); st3(c.local_limit_bitangent_a, bt);
            st3(c.local_limit_tangent_b, conjugate(B.rotation) * (A.rotation * t));
            c.swing_motor_type = MI_MOTOR_VELOCITY; c.swing_motor_velocity_or_target_angle = 0.f; c.max_swing_motor_torque = -1.f; c.swing_motor_axis = 0.f;
            c.twist_motor_type = MI_MOTOR_VELOCITY; c.twist_motor_velocity_or_target_angle = 0.f; c.max_twist_motor_torque = -1.f;
            return jointsAdd(w, type, ea, eb, &c, sizeof(c), out);
        }
        case MI_CONSTRAINT_SLIDER: {
            mi_slider_constraint c;
            st3(c.local_anchor_a, invPos(A, ga)); st3(c.local_anchor_b, invPos(B, ga));
            st3(c.local_axis_a, invDir(A, gx));
            st4(c.initial_inv_rotation_difference, conjugate(B.rotation) * A.rotation);
            c.neg_distance_limit = l0; c.pos_distance_limit = l1;
            c.motor_type = MI_MOTOR_VELOCITY; c.motor_velocity_or_target_distance = 0.f; c.max_motor_force = -1.f;
            return jointsAdd(w, type, ea, eb, &c, sizeof(c), out);
        }
    }
    return MI_ERR_INVALID_ARGUMENT;
}

// Solve order per type: reference mode = creation order; canonical mode = greedy colouring (descending hash32
// priority, lowest colour free on both dynamic bodies, 64 colours + overflow), colour-major / index-minor.
static void computeOrder(const World& w, const std::vector<Pair>& bodies, std::vector<uint32_t>& order) {
    uint32_t n = (uint32_t)bodies.size();
    order.resize(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    if (w.orderMode == 0 || w.debugOrderPending || n == 0) return;   // (ora_debug_set_solve_order: one step in the reference's order)
    std::vector<uint32_t> prio(order);
    std::sort(prio.begin(), prio.end(), [](uint32_t a, uint32_t b) { return hash32(a) > hash32(b); });
    std::vector<uint64_t> used(w.bodies.size(), 0);
    std::vector<uint32_t> color(n, 64);
    for (uint32_t j : prio) {
        Pair bp = bodies[j];
        bool dynA = w.bodies[bp.a].invMass != 0.f, dynB = w.bodies[bp.b].invMass != 0.f;
        uint64_t mask = (dynA ? used[bp.a] : 0) | (dynB ? used[bp.b] : 0);
        if (~mask == 0) continue;
        uint32_t c = (uint32_t)__builtin_ctzll(~mask);
        color[j] = c;
        if (dynA) used[bp.a] |= 1ull << c;
        if (dynB) used[bp.b] |= 1ull << c;
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return color[a] < color[b]; });
}

static mat3 ballInvEffMass(const GlobalState& A, const GlobalState& B, vec3 rA, vec3 rB) {
    mat3 sA = getSkewMatrix(rA), sB = getSkewMatrix(rB);
    return sA * A.invInertia * transpose(sA) + sB * B.invInertia * transpose(sB) + mat3::identity() * (A.invMass + B.invMass);
}
static inline float inv0(float x) { return (x != 0.f) ? (1.f / x) : 0.f; }

void jointsMarkOrderDirty(World& w) { w.joints->orderDirty = true; }
void jointsInitialize(World& w, float dt) {
    JointStore& J = *w.joints;
    if (J.orderDirty) {
        computeOrder(w, J.distance.bodies, J.distance.order); computeOrder(w, J.ball.bodies, J.ball.order); computeOrder(w, J.fixed.bodies, J.fixed.order);
        computeOrder(w, J.hinge.bodies, J.hinge.order); computeOrder(w, J.cone.bodies, J.cone.order); computeOrder(w, J.slider.bodies, J.slider.order);
        J.orderDirty = false;
    }
    float invDt = 1.f / dt;
    const std::vector<GlobalState>& rb = w.rb;

    J.uDistance.resize(J.distance.pods.size());
    for (size_t i = 0; i < J.distance.pods.size(); ++i) {  // constraints.cpp:189-237
        const mi_distance_constraint& in = J.distance.pods[i]; DistanceUpd& o = J.uDistance[i];
        o.a = J.distance.bodies[i].a; o.b = J.distance.bodies[i].b;
        const GlobalState& A = rb[o.a]; const GlobalState& B = rb[o.b];
        if (w.shard.enabled && !w.shard.active[o.a]) continue;
        o.rA = A.rotation * (v3(in.local_anchor_a) - A.localCOG);
        o.rB = B.rotation * (v3(in.local_anchor_b) - B.localCOG);
        vec3 gA = A.position + o.rA, gB = B.position + o.rB;
        o.u = gB - gA;
        float l = length(o.u);
        o.u = (l > 0.001f) ? (o.u * (1.f / l)) : vec3(0.f);
        vec3 crAu = cross(o.rA, o.u), crBu = cross(o.rB, o.u);
        float invMass = A.invMass + dot(crAu, A.invInertia * crAu) + B.invMass + dot(crBu, B.invInertia * crBu);
        o.effMass = inv0(invMass);
        o.bias = 0.f;
        if (dt > DT_THRESHOLD) o.bias = (l - in.global_length) * (BETA_DISTANCE * invDt);
        o.iwA = A.invInertia * cross(o.rA, crAu);
        o.iwB = B.invInertia * cross(o.rB, crBu);
    }
    J.uBall.resize(J.ball.pods.size());
    for (size_t i = 0; i < J.ball.pods.size(); ++i) {  // 460-503
        const mi_ball_constraint& in = J.ball.pods[i]; BallUpd& o = J.uBall[i];
        o.a = J.ball.bodies[i].a; o.b = J.ball.bodies[i].b;
        const GlobalState& A = rb[o.a]; const GlobalState& B = rb[o.b];
        if (w.shard.enabled && !w.shard.active[o.a]) continue;
        o.rA = A.rotation * (v3(in.local_anchor_a) - A.localCOG);
        o.rB = B.rotation * (v3(in.local_anchor_b) - B.localCOG);
        vec3 gA = A.position + o.rA, gB = B.position + o.rB;
        o.invEffMass = ballInvEffMass(A, B, o.rA, o.rB);
        o.bias = vec3(0.f);
        if (dt > DT_THRESHOLD) o.bias = (gB - gA) * (BETA_BALL * invDt);
    }
    J.uFixed.resize(J.fixed.pods.size());
    for (size_t i = 0; i < J.fixed.pods.size(); ++i) {  // 736-787
        const mi_fixed_constraint& in = J.fixed.pods[i]; FixedUpd& o = J.uFixed[i];
        o.a = J.fixed.bodies[i].a; o.b = J.fixed.bodies[i].b;
        const GlobalState& A = rb[o.a]; const GlobalState& B = rb[o.b];
        if (w.shard.enabled && !w.shard.active[o.a]) continue;
        o.rA = A.rotation * (v3(in.local_anchor_a) - A.localCOG);
        o.rB = B.rotation * (v3(in.local_anchor_b) - B.localCOG);
        vec3 gA = A.position + o.rA, gB = B.position + o.rB;
        o.invEffT = ballInvEffMass(A, B, o.rA, o.rB);
        o.invEffR = A.invInertia + B.invInertia;
        o.tBias = vec3(0.f); o.rBias = vec3(0.f);
        if (dt > DT_THRESHOLD) {
            o.tBias = (gB - gA) * (BETA_BALL * invDt);
            quat err = B.rotation * q4(in.initial_inv_rotation_difference) * conjugate(A.rotation);
            o.rBias = err.v() * (BETA_SLIDER * invDt * 2.f);
        }
    }
    J.uHinge.resize(J.hinge.pods.size());
    for (size_t i = 0; i < J.hinge.pods.size(); ++i) {  // 1079-1211
        const mi_hinge_constraint& in = J.hinge.pods[i]; HingeUpd& o = J.uHinge[i];
        o.a = J.hinge.bodies[i].a; o.b = J.hinge.bodies[i].b;
        const GlobalState& A = rb[o.a]; const GlobalState& B = rb[o.b];
        if (w.shard.enabled && !w.shard.active[o.a]) continue;
        o.rA = A.rotation * (v3(in.local_anchor_a) - A.localCOG);
        o.rB = B.rotation * (v3(in.local_anchor_b) - B.localCOG);
        vec3 gA = A.position + o.rA, gB = B.position + o.rB;
        o.invEffT = ballInvEffMass(A, B, o.rA, o.rB);
        o.tBias = vec3(0.f);
        if (dt > DT_THRESHOLD) o.tBias = (gB - gA) * (BETA_BALL * invDt);
        vec3 axA = A.rotation * v3(in.local_hinge_axis_a), axB = B.rotation * v3(in.local_hinge_axis_b);
        vec3 tB, btB; getTangents(axB, tB, btB);
        vec3 bxa = cross(tB, axA), cxa = cross(btB, axA);
        vec3 iAbxa = A.invInertia * bxa, iBbxa = B.invInertia * bxa, iAcxa = A.invInertia * cxa, iBcxa = B.invInertia * cxa;
        o.invEffR.m00 = dot(bxa, iAbxa) + dot(bxa, iBbxa);
        o.invEffR.m01 = dot(bxa, iAcxa) + dot(bxa, iBcxa);
        o.invEffR.m10 = dot(cxa, iAbxa) + dot(cxa, iBbxa);
        o.invEffR.m11 = dot(cxa, iAcxa) + dot(cxa, iBcxa);
        o.bxa = bxa; o.cxa = cxa;
        o.rBias = vec2{0.f, 0.f};
        if (dt > DT_THRESHOLD) { float k = BETA_HINGE_ROT * invDt; o.rBias = vec2{dot(axA, tB) * k, dot(axA, btB) * k}; }
        o.solveLimit = false; o.solveMotor = false; o.axis = vec3(0.f);
        o.limitImpulse = o.effAxial = o.limitSign = o.maxMotorImpulse = o.motorImpulse = o.motorVelocity = o.limitBias = 0.f;
        o.mlA = o.mlB = vec3(0.f);
        if (in.min_rotation_limit <= 0.f || in.max_rotation_limit >= 0.f || in.max_motor_torque > 0.f) {
            vec3 cmp = conjugate(A.rotation) * (B.rotation * v3(in.local_hinge_tangent_b));
            float angle = det_atan2f(dot(cmp, v3(in.local_hinge_bitangent_a)), dot(cmp, v3(in.local_hinge_tangent_a)));
            bool minV = in.min_rotation_limit <= 0.f && angle <= in.min_rotation_limit;
            bool maxV = in.max_rotation_limit >= 0.f && angle >= in.max_rotation_limit;
            o.solveLimit = minV || maxV;
            o.solveMotor = in.max_motor_torque > 0.f;
            if (o.solveLimit || o.solveMotor) {
                o.axis = axA;
                o.limitImpulse = 0.f;
                float invAx = dot(axA, A.invInertia * axA) + dot(axA, B.invInertia * axA);
                o.effAxial = inv0(invAx);
                o.limitSign = minV ? 1.f : -1.f;
                o.maxMotorImpulse = in.max_motor_torque * dt;
                o.motorImpulse = 0.f;
                o.mlA = A.invInertia * o.axis; o.mlB = B.invInertia * o.axis;
                o.motorVelocity = in.motor_velocity_or_target_angle;
                if (in.motor_type == MI_MOTOR_POSITION) {
                    float minL = (in.min_rotation_limit <= 0.f) ? in.min_rotation_limit : -kPi;
                    float maxL = (in.max_rotation_limit >= 0.f) ? in.max_rotation_limit : kPi;
                    float target = clampf(in.motor_velocity_or_target_angle, minL, maxL);
                    o.motorVelocity = (dt > DT_THRESHOLD) ? ((target - angle) * invDt) : 0.f;
                }
                o.limitBias = 0.f;
                if (dt > DT_THRESHOLD) {
                    float d = minV ? (angle - in.min_rotation_limit) : (in.max_rotation_limit - angle);
                    o.limitBias = d * BETA_HINGE_LIMIT * invDt;
                }
            }
        }
    }
    J.uCone.resize(J.cone.pods.size());
    for (size_t i = 0; i < J.cone.pods.size(); ++i) {  // 1782-1950
        const mi_cone_twist_constraint& in = J.cone.pods[i]; ConeUpd& o = J.uCone[i];
        std::memset((void*)&o, 0, sizeof(o));
        o.a = J.cone.bodies[i].a; o.b = J.cone.bodies[i].b;
        const GlobalState& A = rb[o.a]; const GlobalState& B = rb[o.b];
        if (w.shard.enabled && !w.shard.active[o.a]) continue;
        o.rA = A.rotation * (v3(in.local_anchor_a) - A.localCOG);
        o.rB = B.rotation * (v3(in.local_anchor_b) - B.localCOG);
        vec3 gA = A.position + o.rA, gB = B.position + o.rB;
        o.invEff = ballInvEffMass(A, B, o.rA, o.rB);
        o.bias = vec3(0.f);
        if (dt > DT_THRESHOLD) o.bias = (gB - gA) * (BETA_BALL * invDt);
        quat btoa = conjugate(A.rotation) * B.rotation;
        vec3 axisA = v3(in.local_limit_axis_a);
        vec3 axisCmpA = btoa * v3(in.local_limit_axis_b);
        quat swingRot = rotateFromTo(axisA, axisCmpA);
        vec3 twistTangentA = swingRot * v3(in.local_limit_tangent_a);
        vec3 twistBitangentA = swingRot * v3(in.local_limit_bitangent_a);
        vec3 tangentCmpA = btoa * v3(in.local_limit_tangent_b);
        float twistAngle = det_atan2f(dot(tangentCmpA, twistBitangentA), dot(tangentCmpA, twistTangentA));
        vec3 swingAxis; float swingAngle;
        getAxisRotation(swingRot, swingAxis, swingAngle);
        if (swingAngle < 0.f) { swingAngle *= -1.f; swingAxis *= -1.f; }
        o.solveSwingLimit = in.swing_limit >= 0.f && swingAngle >= in.swing_limit;
        if (o.solveSwingLimit) {
            o.swingImpulse = 0.f;
            o.swingAxis = A.rotation * swingAxis;
            float invM = dot(o.swingAxis, A.invInertia * o.swingAxis) + dot(o.swingAxis, B.invInertia * o.swingAxis);
            o.effSwingLimit = inv0(invM);
            o.swingLimitBias = 0.f;
            if (dt > DT_THRESHOLD) o.swingLimitBias = (in.swing_limit - swingAngle) * (BETA_HINGE_LIMIT * invDt);
            o.slA = A.invInertia * o.swingAxis; o.slB = B.invInertia * o.swingAxis;
        }
        o.solveSwingMotor = in.max_swing_motor_torque > 0.f;
        if (o.solveSwingMotor) {
            o.maxSwingMotorImpulse = in.max_swing_motor_torque * dt;
            o.swingMotorImpulse = 0.f;
            float axisX = det_cosf(in.swing_motor_axis), axisY = det_sinf(in.swing_motor_axis);
            vec3 localMotorAxis = axisX * v3(in.local_limit_tangent_a) + axisY * v3(in.local_limit_bitangent_a);
            if (in.swing_motor_type == MI_MOTOR_VELOCITY) {
                o.swingMotorAxis = A.rotation * localMotorAxis;
                o.swingMotorVelocity = in.swing_motor_velocity_or_target_angle;
            } else {
                float target = in.swing_motor_velocity_or_target_angle;
                if (in.swing_limit >= 0.f) target = clampf(target, -in.swing_limit, in.swing_limit);
                float h = target * 0.5f;   // quat(axis, angle): w = cos(angle/2), v = axis * sin(angle/2)  (src/core/math.h:932-936)
                float sh = det_sinf(h), ch = det_cosf(h);
                quat tq(localMotorAxis.x * sh, localMotorAxis.y * sh, localMotorAxis.z * sh, ch);
                vec3 localTargetDir = tq * axisA;
                vec3 localMotorAxis2 = noz(cross(axisCmpA, localTargetDir));
                o.swingMotorAxis = A.rotation * localMotorAxis2;
                float cosAngle = dot(localTargetDir, axisCmpA);
                float deltaAngle = det_acosf(clamp01(cosAngle));
                o.swingMotorVelocity = (dt > DT_THRESHOLD) ? (deltaAngle * invDt * 0.2f) : 0.f;
            }
            o.smA = A.invInertia * o.swingMotorAxis; o.smB = B.invInertia * o.swingMotorAxis;
            float invM = dot(o.swingMotorAxis, A.invInertia * o.swingMotorAxis) + dot(o.swingMotorAxis, B.invInertia * o.swingMotorAxis);
            o.effSwingMotor = inv0(invM);
        }
        bool minTw = in.twist_limit >= 0.f && twistAngle <= -in.twist_limit;
        bool maxTw = in.twist_limit >= 0.f && twistAngle >= in.twist_limit;
        o.solveTwistLimit = minTw || maxTw;
        o.solveTwistMotor = in.max_twist_motor_torque > 0.f;
        if (o.solveTwistLimit || o.solveTwistMotor) {
            o.twistImpulse = 0.f;
            o.twistAxis = A.rotation * axisA;
            float invM = dot(o.twistAxis, A.invInertia * o.twistAxis) + dot(o.twistAxis, B.invInertia * o.twistAxis);
            o.effTwist = inv0(invM);
            o.twistLimitSign = minTw ? 1.f : -1.f;
            o.maxTwistMotorImpulse = in.max_twist_motor_torque * dt;
            o.twistMotorImpulse = 0.f;
            o.tmA = A.invInertia * o.twistAxis; o.tmB = B.invInertia * o.twistAxis;
            o.twistMotorVelocity = in.twist_motor_velocity_or_target_angle;
            if (in.twist_motor_type == MI_MOTOR_POSITION) {
                float limit = (in.twist_limit >= 0.f) ? in.twist_limit : kPi;
                float target = clampf(in.twist_motor_velocity_or_target_angle, -limit, limit);
                o.twistMotorVelocity = (dt > DT_THRESHOLD) ? ((target - twistAngle) * invDt) : 0.f;
            }
            o.twistLimitBias = 0.f;
            if (dt > DT_THRESHOLD) {
                float d = minTw ? (in.twist_limit + twistAngle) : (in.twist_limit - twistAngle);
                o.twistLimitBias = d * BETA_TWIST_LIMIT * invDt;
            }
        }
    }
    J.uSlider.resize(J.slider.pods.size());
    for (size_t i = 0; i < J.slider.pods.size(); ++i) {  // 2638-2762
        const mi_slider_constraint& in = J.slider.pods[i]; SliderUpd& o = J.uSlider[i];
        std::memset((void*)&o, 0, sizeof(o));
        o.a = J.slider.bodies[i].a; o.b = J.slider.bodies[i].b;
        const GlobalState& A = rb[o.a]; const GlobalState& B = rb[o.b];
        if (w.shard.enabled && !w.shard.active[o.a]) continue;
        vec3 rA = A.rotation * (v3(in.local_anchor_a) - A.localCOG);
        vec3 rB = B.rotation * (v3(in.local_anchor_b) - B.localCOG);
        vec3 gA = A.position + rA, gB = B.position + rB;
        vec3 axis = A.rotation * v3(in.local_axis_a);
        getTangents(axis, o.tangent, o.bitangent);
        vec3 u = gB - gA;
        vec3 rAu = rA + u;
        o.rBxt = cross(rB, o.tangent); o.rBxb = cross(rB, o.bitangent);
        o.rAuxt = cross(rAu, o.tangent); o.rAuxb = cross(rAu, o.bitangent);
        vec3 iArAuxt = A.invInertia * o.rAuxt, iArAuxb = A.invInertia * o.rAuxb, iBrBxt = B.invInertia * o.rBxt, iBrBxb = B.invInertia * o.rBxb;
        float invMassSum = A.invMass + B.invMass;
        o.invEffT.m00 = dot(o.rAuxt, iArAuxt) + dot(o.rBxt, iBrBxt) + invMassSum;
        o.invEffT.m01 = dot(o.rAuxt, iArAuxb) + dot(o.rBxt, iBrBxb);
        o.invEffT.m10 = dot(o.rAuxb, iArAuxt) + dot(o.rBxb, iBrBxt);
        o.invEffT.m11 = dot(o.rAuxb, iArAuxb) + dot(o.rBxb, iBrBxb) + invMassSum;
        o.invEffR = A.invInertia + B.invInertia;
        o.tBias = vec2{0.f, 0.f}; o.rBias = vec3(0.f);
        if (dt > DT_THRESHOLD) {
            float a = dot(u, o.tangent), b = dot(u, o.bitangent);
            float k = BETA_SLIDER * invDt;
            o.tBias = vec2{a * k, b * k};
            quat err = B.rotation * q4(in.initial_inv_rotation_difference) * conjugate(A.rotation);
            o.rBias = err.v() * (BETA_SLIDER * invDt * 2.f);
        }
        o.axis = axis;
        float dist = dot(u, axis);
        o.solveLimit = false;
        if (in.neg_distance_limit <= 0.f || in.pos_distance_limit >= 0.f) {
            bool minV = (in.neg_distance_limit <= 0.f) && (dist < in.neg_distance_limit);
            bool maxV = (in.pos_distance_limit >= 0.f) && (dist > in.pos_distance_limit);
            if (minV || maxV) {
                o.solveLimit = true;
                o.limitImpulse = 0.f;
                o.rAuxs = cross(rAu, axis); o.rBxs = cross(rB, axis);
                float invAx = invMassSum + dot(o.rAuxs, A.invInertia * o.rAuxs) + dot(o.rBxs, B.invInertia * o.rBxs);
                o.effAxial = inv0(invAx);
                o.limitSign = minV ? 1.f : -1.f;
                o.limitBias = 0.f;
                if (dt > DT_THRESHOLD) {
                    float err = minV ? (dist - in.neg_distance_limit) : (in.pos_distance_limit - dist);
                    o.limitBias = err * (BETA_SLIDER_LIMIT * invDt);
                }
                o.llA = A.invInertia * o.rAuxs; o.llB = B.invInertia * o.rBxs;
            }
        }
        o.solveMotor = false;
        if (in.max_motor_force > 0.f) {
            o.solveMotor = true;
            o.maxMotorImpulse = in.max_motor_force * dt;
            o.motorImpulse = 0.f;
            o.motorVelocity = in.motor_velocity_or_target_distance;
            if (in.motor_type == MI_MOTOR_POSITION) {
                float minL = (in.neg_distance_limit <= 0.f) ? in.neg_distance_limit : -INFINITY;
                float maxL = (in.pos_distance_limit >= 0.f) ? in.pos_distance_limit : INFINITY;
                float target = clampf(in.motor_velocity_or_target_distance, minL, maxL);
                o.motorVelocity = (dt > DT_THRESHOLD) ? ((target - dist) * invDt) : 0.f;
            }
        }
    }
}

static inline vec2 add2(vec2 a, vec2 b) { return vec2{a.x + b.x, a.y + b.y}; }

void jointsSolveIteration(World& w) {
    JointStore& J = *w.joints;
    std::vector<GlobalState>& rb = w.rb;
    for (uint32_t i : J.distance.order) {  // 239-264
        DistanceUpd& c = J.uDistance[i]; GlobalState& A = rb[c.a]; GlobalState& B = rb[c.b];
        if (w.shard.enabled && !w.shard.active[c.a]) continue;   // sharded world: an island this rank does not simulate
        vec3 avA = A.linearVelocity + cross(A.angularVelocity, c.rA);
        vec3 avB = B.linearVelocity + cross(B.angularVelocity, c.rB);
        float Cdot = dot(c.u, avB - avA) + c.bias;
        float lambda = -c.effMass * Cdot;
        vec3 P = lambda * c.u;
        A.linearVelocity -= A.invMass * P;
        A.angularVelocity -= c.iwA * lambda;
        B.linearVelocity += B.invMass * P;
        B.angularVelocity += c.iwB * lambda;
    }
    for (uint32_t i : J.ball.order) {  // 505-528
        BallUpd& c = J.uBall[i]; GlobalState& A = rb[c.a]; GlobalState& B = rb[c.b];
        if (w.shard.enabled && !w.shard.active[c.a]) continue;   // sharded world: an island this rank does not simulate
        vec3 avA = A.linearVelocity + cross(A.angularVelocity, c.rA);
        vec3 avB = B.linearVelocity + cross(B.angularVelocity, c.rB);
        vec3 Cdot = avB - avA + c.bias;
        vec3 P = solveLinearSystem(c.invEffMass, -Cdot);
        A.linearVelocity -= A.invMass * P;
        A.angularVelocity -= A.invInertia * cross(c.rA, P);
        B.linearVelocity += B.invMass * P;
        B.angularVelocity += B.invInertia * cross(c.rB, P);
    }
    for (uint32_t i : J.fixed.order) {  // 789-823
        FixedUpd& c = J.uFixed[i]; GlobalState& A = rb[c.a]; GlobalState& B = rb[c.b];
        if (w.shard.enabled && !w.shard.active[c.a]) continue;   // sharded world: an island this rank does not simulate
        {
            vec3 Cdot = B.angularVelocity - A.angularVelocity;
            vec3 rl = solveLinearSystem(c.invEffR, -(Cdot + c.rBias));
            A.angularVelocity -= A.invInertia * rl;
            B.angularVelocity += B.invInertia * rl;
        }
        {
            vec3 avA = A.linearVelocity + cross(A.angularVelocity, c.rA);
            vec3 avB = B.linearVelocity + cross(B.angularVelocity, c.rB);
            vec3 Cdot = avB - avA + c.tBias;
            vec3 P = solveLinearSystem(c.invEffT, -Cdot);
            A.linearVelocity -= A.invMass * P;
            A.angularVelocity -= A.invInertia * cross(c.rA, P);
            B.linearVelocity += B.invMass * P;
            B.angularVelocity += B.invInertia * cross(c.rB, P);
        }
    }
    for (uint32_t i : J.hinge.order) {  // 1213-1307
        HingeUpd& c = J.uHinge[i]; GlobalState& A = rb[c.a]; GlobalState& B = rb[c.b];
        if (w.shard.enabled && !w.shard.active[c.a]) continue;   // sharded world: an island this rank does not simulate
        vec3 vA = A.linearVelocity, wA = A.angularVelocity, vB = B.linearVelocity, wB = B.angularVelocity;
        vec3 axis = c.axis;
        if (c.solveMotor) {
            float aA = dot(axis, wA), aB = dot(axis, wB);
            float rel = (aB - aA);
            float cd = rel - c.motorVelocity;
            float l = -c.effAxial * cd;
            float old = c.motorImpulse;
            c.motorImpulse = clampf(c.motorImpulse + l, -c.maxMotorImpulse, c.maxMotorImpulse);
            l = c.motorImpulse - old;
            wA -= c.mlA * l;
            wB += c.mlB * l;
        }
        if (c.solveLimit) {
            float s = c.limitSign;
            float aA = dot(axis, wA), aB = dot(axis, wB);
            float rel = s * (aB - aA);
            float cd = rel + c.limitBias;
            float l = -c.effAxial * cd;
            float imp = fmax2(c.limitImpulse + l, 0.f);
            l = imp - c.limitImpulse;
            c.limitImpulse = imp;
            l *= s;
            wA -= c.mlA * l;
            wB += c.mlB * l;
        }
        {
            vec3 dw = wB - wA;
            vec2 cd{dot(c.bxa, dw), dot(c.cxa, dw)};
            vec2 s = add2(cd, c.rBias);
            vec2 rl = solve2(c.invEffR, vec2{-s.x, -s.y});
            vec3 P = c.bxa * rl.x + c.cxa * rl.y;
            wA -= A.invInertia * P;
            wB += B.invInertia * P;
        }
        {
            vec3 avA = vA + cross(wA, c.rA);
            vec3 avB = vB + cross(wB, c.rB);
            vec3 cd = avB - avA + c.tBias;
            vec3 P = solveLinearSystem(c.invEffT, -cd);
            vA -= A.invMass * P;
            wA -= A.invInertia * cross(c.rA, P);
            vB += B.invMass * P;
            wB += B.invInertia * cross(c.rB, P);
        }
        A.linearVelocity = vA; A.angularVelocity = wA; B.linearVelocity = vB; B.angularVelocity = wB;
    }
    for (uint32_t i : J.cone.order) {  // 1952-2070
        ConeUpd& c = J.uCone[i]; GlobalState& A = rb[c.a]; GlobalState& B = rb[c.b];
        if (w.shard.enabled && !w.shard.active[c.a]) continue;   // sharded world: an island this rank does not simulate
        vec3 vA = A.linearVelocity, wA = A.angularVelocity, vB = B.linearVelocity, wB = B.angularVelocity;
        vec3 tw = c.twistAxis;
        if (c.solveTwistMotor) {
            float aA = dot(tw, wA), aB = dot(tw, wB);
            float rel = (aB - aA);
            float cd = rel - c.twistMotorVelocity;
            float l = -c.effTwist * cd;
            float old = c.twistMotorImpulse;
            c.twistMotorImpulse = clampf(c.twistMotorImpulse + l, -c.maxTwistMotorImpulse, c.maxTwistMotorImpulse);
            l = c.twistMotorImpulse - old;
            wA -= c.tmA * l;
            wB += c.tmB * l;
        }
        if (c.solveSwingMotor) {
            vec3 ax = c.swingMotorAxis;
            float aA = dot(ax, wA), aB = dot(ax, wB);
            float rel = (aB - aA);
            float cd = rel - c.swingMotorVelocity;
            float l = -c.effSwingMotor * cd;
            float old = c.swingMotorImpulse;
            c.swingMotorImpulse = clampf(c.swingMotorImpulse + l, -c.maxSwingMotorImpulse, c.maxSwingMotorImpulse);
            l = c.swingMotorImpulse - old;
            wA -= c.smA * l;
            wB += c.smB * l;
        }
        if (c.solveTwistLimit) {
            float s = c.twistLimitSign;
            float aA = dot(tw, wA), aB = dot(tw, wB);
            float rel = s * (aB - aA);
            float cd = rel + c.twistLimitBias;
            float l = -c.effTwist * cd;
            float imp = fmax2(c.twistImpulse + l, 0.f);
            l = imp - c.twistImpulse;
            c.twistImpulse = imp;
            l *= s;
            wA -= c.tmA * l;
            wB += c.tmB * l;
        }
        if (c.solveSwingLimit) {
            float aA = dot(c.swingAxis, wA), aB = dot(c.swingAxis, wB);
            float cd = aA - aB + c.swingLimitBias;
            float l = -c.effSwingLimit * cd;
            float imp = fmax2(c.swingImpulse + l, 0.f);
            l = imp - c.swingImpulse;
            c.swingImpulse = imp;
            wA += c.slA * l;
            wB -= c.slB * l;
        }
        {
            vec3 avA = vA + cross(wA, c.rA);
            vec3 avB = vB + cross(wB, c.rB);
            vec3 cd = avB - avA + c.bias;
            vec3 P = solveLinearSystem(c.invEff, -cd);
            vA -= A.invMass * P;
            wA -= A.invInertia * cross(c.rA, P);
            vB += B.invMass * P;
            wB += B.invInertia * cross(c.rB, P);
        }
        A.linearVelocity = vA; A.angularVelocity = wA; B.linearVelocity = vB; B.angularVelocity = wB;
    }
    for (uint32_t i : J.slider.order) {  // 2764-2846
        SliderUpd& c = J.uSlider[i]; GlobalState& A = rb[c.a]; GlobalState& B = rb[c.b];
        if (w.shard.enabled && !w.shard.active[c.a]) continue;   // sharded world: an island this rank does not simulate
        vec3 vA = A.linearVelocity, wA = A.angularVelocity, vB = B.linearVelocity, wB = B.angularVelocity;
        if (c.solveMotor) {
            float cd = dot(vB, c.axis) - dot(vA, c.axis) - c.motorVelocity;
            float mass = 1.f / (A.invMass + B.invMass);
            float l = -mass * cd;
            float old = c.motorImpulse;
            c.motorImpulse = clampf(c.motorImpulse + l, -c.maxMotorImpulse, c.maxMotorImpulse);
            l = c.motorImpulse - old;
            vec3 P = l * c.axis;
            vA -= A.invMass * P;
            vB += B.invMass * P;
        }
        if (c.solveLimit) {
            float cd = dot(vB, c.axis) + dot(wB, c.rBxs) - dot(vA, c.axis) - dot(wA, c.rAuxs);
            float l = -c.effAxial * (c.limitSign * cd + c.limitBias);
            float imp = fmax2(c.limitImpulse + l, 0.f);
            l = imp - c.limitImpulse;
            c.limitImpulse = imp;
            l *= c.limitSign;
            vec3 P = l * c.axis;
            vA -= A.invMass * P;
            wA -= c.llA * l;
            vB += B.invMass * P;
            wB += c.llB * l;
        }
        {
            vec3 cd = wB - wA;
            vec3 rl = solveLinearSystem(c.invEffR, -(cd + c.rBias));
            wA -= A.invInertia * rl;
            wB += B.invInertia * rl;
        }
        {
            vec2 cd;
            cd.x = dot(c.tangent, vB) + dot(c.rBxt, wB) - dot(c.tangent, vA) - dot(c.rAuxt, wA);
            cd.y = dot(c.bitangent, vB) + dot(c.rBxb, wB) - dot(c.bitangent, vA) - dot(c.rAuxb, wA);
            vec2 s = add2(cd, c.tBias);
            vec2 tl = solve2(c.invEffT, vec2{-s.x, -s.y});
            vec3 tb = c.tangent * tl.x + c.bitangent * tl.y;
            vA -= A.invMass * tb;
            wA -= A.invInertia * (c.rAuxt * tl.x + c.rAuxb * tl.y);
            vB += B.invMass * tb;
            wB += B.invInertia * (c.rBxt * tl.x + c.rBxb * tl.y);
        }
        A.linearVelocity = vA; A.angularVelocity = wA; B.linearVelocity = vB; B.angularVelocity = wB;
    }
}

}  // namespace ora
