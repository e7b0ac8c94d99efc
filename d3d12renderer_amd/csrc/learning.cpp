// learning.cpp — libPhysics-Lib.so: the reference's learning DLL C ABI over the MI355X physics library.
//
// Replaces src/learning/learned_locomotion.cpp:395-489 (the five __declspec(dllexport) functions consumed by
// learning/loco_env.py through ctypes): getPhysicsStateSize, getPhysicsActionSize, getPhysicsRanges, resetPhysics,
// updatePhysics — same signatures, same state / action layout (learned_locomotion.h:15-65), same reward
// (learned_locomotion.cpp:318-345), same random pushes (458-468), the humanoid of src/physics/ragdoll.cpp:9-123 on the
// reference's ground slab.  Because the physics world behind it is a batch, the same library steps thousands of
// independent environments per call (resetPhysicsBatch / updatePhysicsBatch): environment e lives at its own origin on its
// own ground slab, entities [15 e, 15 e + 15), rays of the random pushes are restricted to that range.
//
// Host side only (no kernels here): it talks to the physics library through the C ABI of include/mi_physics.h (a test build
// may put another implementation of those calls behind the same names: "The physics backend" below).
//
// Stated deviations: resetPhysics also WRITES the initial state to outState (the reference leaves the buffer untouched);
// the push RNG is seeded with a fixed default instead of time(0) (setPhysicsSeed changes it); an episode reset puts the
// ragdoll back in place instead of rebuilding the scene (the contact-colour history of the world survives the reset).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi_physics.h"
#include "../../include/mi_constraints.h"

// The physics backend.  The product build talks to libmi_physics.so through include/mi_physics.h.  A build may supply another implementation of the same calls through
// a header of its own (-DMI_LEARNING_BACKEND_HEADER='"path"'; the parity tests do, to run this environment code over their checker): that header defines PHYS(name),
// phys_world, physCreateWorld(device, out) and physLastError().  Nothing here knows what such a backend is.
#ifdef MI_LEARNING_BACKEND_HEADER
#include MI_LEARNING_BACKEND_HEADER
#else
#define PHYS(name) mi_##name
typedef mi_world phys_world;
static inline int physCreateWorld(int device, phys_world** out) { mi_world_desc desc{}; desc.device = device; return mi_world_create(&desc, out); }
static inline const char* physLastError() { return mi_last_error(); }
#endif

#define EXPORT extern "C" __attribute__((visibility("default")))

namespace {

// ---- the little math the environment needs (src/core/math.h; operation order kept) ---------------------------------------
struct v3 { float x, y, z; };
struct q4 { float x, y, z, w; };
inline v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline v3 operator*(float s, v3 a) { return a * s; }
inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline v3 cross(v3 a, v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(v3 a) { return std::sqrt(dot(a, a)); }
inline v3 normalize(v3 a) { float l = length(a); return a * (1.f / l); }
inline q4 conjugate(q4 a) { return {-a.x, -a.y, -a.z, a.w}; }
inline q4 operator*(q4 a, q4 b) {   // math.h:627-633
    v3 av{a.x, a.y, a.z}, bv{b.x, b.y, b.z};
    float w = a.w * b.w - dot(av, bv);
    v3 v = av * b.w + bv * a.w + cross(av, bv);
    return {v.x, v.y, v.z, w};
}
inline v3 operator*(q4 q, v3 v) { q4 p{v.x, v.y, v.z, 0.f}; q4 r = q * p * conjugate(q); return {r.x, r.y, r.z}; }   // math.h:642-646
inline q4 axisAngle(v3 axis, float angle) {   // quat(vec3 axis, float angle), math.h:335-343
    float h = angle * 0.5f, s = std::sin(h);
    return {axis.x * s, axis.y * s, axis.z * s, std::cos(h)};
}
inline float lerpf(float l, float u, float t) { return l + t * (u - l); }
inline float clampf(float v, float l, float u) { float r = l > v ? l : v; return u < r ? u : r; }
constexpr float kPi = 3.14159265359f;
inline float deg2rad(float d) { return d * (kPi / 180.f); }

// random_number_generator (src/core/random.h:5-52): xorshift64
struct Rng {
    uint64_t state;
    uint64_t next64() { uint64_t x = state; x ^= x << 13; x ^= x >> 7; x ^= x << 17; state = x; return x; }
    uint32_t next32() { return (uint32_t)next64(); }
    uint32_t between(uint32_t lo, uint32_t hi) { return next32() % (hi - lo) + lo; }
    float float01() { return (float)next32() / (float)0xFFFFFFFFu; }
    float floatBetween(float lo, float hi) { return lerpf(lo, hi, (float01() - 0.f) / (1.f - 0.f)); }
};

// ---- the humanoid (src/physics/ragdoll.cpp:9-123) ------------------------------------------------------------------------
constexpr int kParts = 14, kCone = 7, kHinge = 6;
constexpr int kActionFloats = kCone * 3 + kHinge;          // learning_action: 27
constexpr int kStateFloats = 13 * 3 + kActionFloats;       // learning_state: 66
constexpr int kEntitiesPerEnv = kParts + 1;                // ground + body parts
constexpr float kScale = 0.42f;
enum Part { TORSO, HEAD, L_UPPER_ARM, L_LOWER_ARM, R_UPPER_ARM, R_LOWER_ARM, L_UPPER_LEG, L_LOWER_LEG, L_FOOT, L_TOES, R_UPPER_LEG, R_LOWER_LEG, R_FOOT, R_TOES };
const int kParent[kParts] = {-1, TORSO, TORSO, L_UPPER_ARM, TORSO, R_UPPER_ARM, TORSO, L_UPPER_LEG, L_LOWER_LEG, L_FOOT, TORSO, R_UPPER_LEG, R_LOWER_LEG, R_FOOT};
struct PartDef { v3 pos; float zDeg; };
const PartDef kPartDefs[kParts] = {
    {{0.f, 0.f, 0.f}, 0.f}, {{0.f, 1.45f, 0.f}, 0.f},
    {{-0.6f, 0.75f, 0.f}, -30.f}, {{-0.884f, 0.044f, -0.043f}, -20.f}, {{0.6f, 0.75f, 0.f}, 30.f}, {{0.884f, 0.044f, -0.043f}, 20.f},
    {{-0.371f, -0.812f, 0.f}, -10.f}, {{-0.452f, -1.955f, 0.f}, -3.5f}, {{-0.498f, -2.585f, -0.18f}, 0.f}, {{-0.498f, -2.585f, -0.637f}, 0.f},
    {{0.371f, -0.812f, 0.f}, 10.f}, {{0.452f, -1.955f, 0.f}, 3.5f}, {{0.498f, -2.585f, -0.18f}, 0.f}, {{0.498f, -2.585f, -0.637f}, 0.f}};
struct ColDef { int part; bool box; v3 a, b; float r; };   // capsule (a, b, r) or AABB (centre a = 0, half extents b), before `scale *`
const ColDef kColDefs[] = {
    {TORSO, false, {-0.2f, 0.f, 0.f}, {0.2f, 0.f, 0.f}, 0.25f}, {TORSO, false, {-0.16f, 0.32f, 0.f}, {0.16f, 0.32f, 0.f}, 0.2f},
    {TORSO, false, {-0.14f, 0.62f, 0.f}, {0.14f, 0.62f, 0.f}, 0.22f}, {TORSO, false, {-0.14f, 0.92f, 0.f}, {0.14f, 0.92f, 0.f}, 0.2f},
    {HEAD, false, {0.f, -0.075f, 0.f}, {0.f, 0.075f, 0.f}, 0.25f},
    {L_UPPER_ARM, false, {0.f, -0.2f, 0.f}, {0.f, 0.2f, 0.f}, 0.15f}, {L_LOWER_ARM, false, {0.f, -0.2f, 0.f}, {0.f, 0.2f, 0.f}, 0.15f},
    {R_UPPER_ARM, false, {0.f, -0.2f, 0.f}, {0.f, 0.2f, 0.f}, 0.15f}, {R_LOWER_ARM, false, {0.f, -0.2f, 0.f}, {0.f, 0.2f, 0.f}, 0.15f},
    {L_UPPER_LEG, false, {0.f, -0.3f, 0.f}, {0.f, 0.3f, 0.f}, 0.25f}, {L_LOWER_LEG, false, {0.f, -0.3f, 0.f}, {0.f, 0.3f, 0.f}, 0.18f},
    {L_FOOT, true, {0.f, 0.f, 0.f}, {0.1587f, 0.1f, 0.3424f}, 0.f}, {L_TOES, false, {-0.0587f, 0.f, 0.f}, {0.0587f, 0.f, 0.f}, 0.1f},
    {R_UPPER_LEG, false, {0.f, -0.3f, 0.f}, {0.f, 0.3f, 0.f}, 0.25f}, {R_LOWER_LEG, false, {0.f, -0.3f, 0.f}, {0.f, 0.3f, 0.f}, 0.18f},
    {R_FOOT, true, {0.f, 0.f, 0.f}, {0.1587f, 0.1f, 0.3424f}, 0.f}, {R_TOES, false, {-0.0587f, 0.f, 0.f}, {0.0587f, 0.f, 0.f}, 0.1f}};
constexpr int kNumCols = sizeof(kColDefs) / sizeof(kColDefs[0]);
// joints in creation order (ragdoll.cpp:100-116); coneIndex / hingeIndex = slot in humanoid_ragdoll::coneTwistConstraints / hingeConstraints
struct JointDef { bool cone; int slot, a, b, anchorPart; v3 anchor; int axisPart; v3 axis; bool normalizeAxis; float l0, l1; };
const JointDef kJointDefs[] = {
    {true, 0, TORSO, HEAD, TORSO, {0.f, 1.2f, 0.f}, -1, {0.f, 1.f, 0.f}, false, 50.f, 90.f},
    {true, 1, TORSO, L_UPPER_ARM, TORSO, {-0.4f, 1.f, 0.f}, -1, {-1.f, 0.f, 0.f}, false, 130.f, 90.f},
    {false, 0, L_UPPER_ARM, L_LOWER_ARM, L_UPPER_ARM, {0.f, -0.42f, 0.f}, -1, {1.f, 0.f, 1.f}, true, -5.f, 85.f},
    {true, 2, TORSO, R_UPPER_ARM, TORSO, {0.4f, 1.f, 0.f}, -1, {1.f, 0.f, 0.f}, false, 130.f, 90.f},
    {false, 1, R_UPPER_ARM, R_LOWER_ARM, R_UPPER_ARM, {0.f, -0.42f, 0.f}, -1, {1.f, 0.f, -1.f}, true, -5.f, 85.f},
    {true, 3, TORSO, L_UPPER_LEG, TORSO, {-0.3f, -0.25f, 0.f}, L_UPPER_LEG, {0.f, -1.f, 0.f}, false, -1000.f, 30.f},
    {false, 2, L_UPPER_LEG, L_LOWER_LEG, L_UPPER_LEG, {0.f, -0.6f, 0.f}, -1, {1.f, 0.f, 0.f}, false, -90.f, 5.f},
    {true, 4, L_LOWER_LEG, L_FOOT, L_LOWER_LEG, {0.f, -0.52f, 0.f}, L_LOWER_LEG, {0.f, -1.f, 0.f}, false, 75.f, 20.f},
    {false, 3, L_FOOT, L_TOES, L_FOOT, {0.f, 0.f, -0.36f}, -1, {1.f, 0.f, 0.f}, false, -45.f, 45.f},
    {true, 5, TORSO, R_UPPER_LEG, TORSO, {0.3f, -0.25f, 0.f}, R_UPPER_LEG, {0.f, -1.f, 0.f}, false, -1000.f, 30.f},
    {false, 4, R_UPPER_LEG, R_LOWER_LEG, R_UPPER_LEG, {0.f, -0.6f, 0.f}, -1, {1.f, 0.f, 0.f}, false, -90.f, 5.f},
    {true, 6, R_LOWER_LEG, R_FOOT, R_LOWER_LEG, {0.f, -0.52f, 0.f}, R_LOWER_LEG, {0.f, -1.f, 0.f}, false, 75.f, 20.f},
    {false, 5, R_FOOT, R_TOES, R_FOOT, {0.f, 0.f, -0.36f}, -1, {1.f, 0.f, 0.f}, false, -45.f, 45.f}};

struct Target { v3 pos[6], vel[6]; q4 localRot; };   // learning_target
struct Env {
    float smoothed[kActionFloats];   // lastSmoothedAction
    float headTargetHeight;
    v3 torsoVelocityTarget;
    v3 localPositions[kParts][6];
    Target targets[kParts];
    Rng rng;
    float totalReward;
    v3 origin;
};

struct Batch {
    phys_world* world = nullptr;
    int n = 0;
    std::vector<Env> envs;
    std::vector<mi_cone_twist_constraint> cones;     // [env][slot]
    std::vector<mi_hinge_constraint> hinges;
    std::vector<uint32_t> coneIds, hingeIds;         // constraint ids, same layout
    std::vector<float> initialStates;                // [env][part][13]
    std::vector<float> pos, rot, lin, ang;           // per entity, refreshed after every step (transform_component, rb velocities)
    v3 localCOG[kParts];
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    int device = 0;
    std::string error;
};
Batch g;
// host threads of the per-environment loops: a fixed small number (MI_LEARN_THREADS overrides) — containers often expose far
// more logical CPUs than their quota allows, and an OpenMP team sized from that count is slower than one thread
const int g_threads = [] { const char* e = std::getenv("MI_LEARN_THREADS"); int t = e ? std::atoi(e) : 8; return t < 1 ? 1 : t; }();

bool ok(int rc, const char* what) {
    if (rc == MI_OK) return true;
    g.error = std::string(what) + " failed with status " + std::to_string(rc);
    if (const char* detail = physLastError()) g.error += std::string(": ") + detail;
    return false;
}
uint32_t entityOf(int env, int part) { return (uint32_t)(env * kEntitiesPerEnv + 1 + part); }
v3 entityPos(uint32_t e) { return {g.pos[3 * e], g.pos[3 * e + 1], g.pos[3 * e + 2]}; }
q4 entityRot(uint32_t e) { return {g.rot[4 * e], g.rot[4 * e + 1], g.rot[4 * e + 2], g.rot[4 * e + 3]}; }
v3 entityLin(uint32_t e) { return {g.lin[3 * e], g.lin[3 * e + 1], g.lin[3 * e + 2]}; }
v3 entityAng(uint32_t e) { return {g.ang[3 * e], g.ang[3 * e + 1], g.ang[3 * e + 2]}; }
v3 globalCOG(uint32_t e, int part) { return entityPos(e) + entityRot(e) * g.localCOG[part]; }   // rigid_body_component::getGlobalCOGPosition

void destroyWorld() {
    if (g.world) PHYS(world_destroy)(g.world);
    g.world = nullptr; g.n = 0;
}

bool refreshTransforms() {
    const uint32_t ne = (uint32_t)(g.n * kEntitiesPerEnv);
    g.pos.resize(3 * (size_t)ne); g.rot.resize(4 * (size_t)ne); g.lin.resize(3 * (size_t)ne); g.ang.resize(3 * (size_t)ne);
    return ok(PHYS(world_get_transforms)(g.world, g.pos.data(), g.rot.data(), ne), "world_get_transforms") &&
           ok(PHYS(world_get_velocities)(g.world, g.lin.data(), g.ang.data(), ne), "world_get_velocities");
}

// training_locomotion::getLocalPositions (learned_locomotion.cpp:190-246): the 6 face centres of the part's local collider AABB
void localPositionsOf(int part, v3 out[6]) {
    v3 mn{FLT_MAX, FLT_MAX, FLT_MAX}, mx{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    auto grow = [&](v3 p) { mn = {std::fmin(mn.x, p.x), std::fmin(mn.y, p.y), std::fmin(mn.z, p.z)}; mx = {std::fmax(mx.x, p.x), std::fmax(mx.y, p.y), std::fmax(mx.z, p.z)}; };
    // the collider list of an entity is newest first; min / max do not care
    for (int c = 0; c < kNumCols; ++c) {
        if (kColDefs[c].part != part) continue;
        const ColDef& d = kColDefs[c];
        if (d.box) { v3 ctr = kScale * d.a, r = kScale * d.b; grow(ctr - r); grow(ctr + r); }
        else {
            float radius = kScale * d.r; v3 r3{radius, radius, radius}; v3 a = kScale * d.a, b = kScale * d.b;
            v3 bmn{FLT_MAX, FLT_MAX, FLT_MAX}, bmx{-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (v3 p : {a + r3, a - r3, b + r3, b - r3}) { bmn = {std::fmin(bmn.x, p.x), std::fmin(bmn.y, p.y), std::fmin(bmn.z, p.z)}; bmx = {std::fmax(bmx.x, p.x), std::fmax(bmx.y, p.y), std::fmax(bmx.z, p.z)}; }
            grow(bmn); grow(bmx);
        }
    }
    v3 c = (mn + mx) * 0.5f, r = (mx - mn) * 0.5f;
    out[0] = c - v3{r.x, 0.f, 0.f}; out[1] = c - v3{0.f, r.y, 0.f}; out[2] = c - v3{0.f, 0.f, r.z};
    out[3] = c + v3{r.x, 0.f, 0.f}; out[4] = c + v3{0.f, r.y, 0.f}; out[5] = c + v3{0.f, 0.f, r.z};
}

// learned_locomotion::updateConstraint x13 over the smoothed action (learned_locomotion.cpp:74-115) -> one batched update per type
bool applyActions(const float* actions /* [n][27] or null = all zero */, const std::vector<int>* only = nullptr) {
    const float beta = 0.1f;
    std::vector<uint8_t> pick;
    if (only) { pick.assign(g.n, 0); for (int e : *only) pick[e] = 1; }
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g.n >= 256)
    for (int e = 0; e < g.n; ++e) {
        if (only && !pick[e]) continue;
        Env& env = g.envs[e];
        for (int i = 0; i < kActionFloats; ++i) env.smoothed[i] = lerpf(env.smoothed[i], actions ? actions[e * kActionFloats + i] : 0.f, beta);
        for (int s = 0; s < kCone; ++s) {
            mi_cone_twist_constraint& c = g.cones[e * kCone + s];
            c.max_swing_motor_torque = 200.f; c.max_twist_motor_torque = 200.f;
            c.swing_motor_type = 1u; c.twist_motor_type = 1u;   // constraint_position_motor
            c.twist_motor_velocity_or_target_angle = env.smoothed[3 * s];        // cone_twist_action: twistTargetAngle, swingTargetAngle, swingAxisAngle
            c.swing_motor_velocity_or_target_angle = env.smoothed[3 * s + 1];
            c.swing_motor_axis = env.smoothed[3 * s + 2];
        }
        for (int s = 0; s < kHinge; ++s) {
            mi_hinge_constraint& h = g.hinges[e * kHinge + s];
            h.max_motor_torque = 200.f; h.motor_type = 1u;
            h.motor_velocity_or_target_angle = env.smoothed[kCone * 3 + s];
        }
    }
    return ok(PHYS(constraints_update)(g.world, MI_CONSTRAINT_CONE_TWIST, (uint32_t)g.coneIds.size(), g.coneIds.data(), g.cones.data(), sizeof(mi_cone_twist_constraint)), "constraints_update") &&
           ok(PHYS(constraints_update)(g.world, MI_CONSTRAINT_HINGE, (uint32_t)g.hingeIds.size(), g.hingeIds.data(), g.hinges.data(), sizeof(mi_hinge_constraint)), "constraints_update");
}

// learned_locomotion::getState (learned_locomotion.cpp:117-156); returns hasFallen (head below 1 m)
bool stateOf(int e, float* out) {
    const Env& env = g.envs[e];
    v3 cog = globalCOG(entityOf(e, TORSO), TORSO);
    cog.y = 0.f;
    auto toLocalPos = [&](v3 p) { return p - cog; };        // trs(cog, identity): conjugate(identity) * (p - cog) / 1
    auto part = [&](int p, float* posOut, float* velOut) {
        v3 lp = toLocalPos(globalCOG(entityOf(e, p), p)), lv = entityLin(entityOf(e, p));
        posOut[0] = lp.x; posOut[1] = lp.y; posOut[2] = lp.z; velOut[0] = lv.x; velOut[1] = lv.y; velOut[2] = lv.z;
    };
    // learning_state layout (learned_locomotion.h:41-65)
    v3 cv = entityLin(entityOf(e, TORSO));
    out[0] = cv.x; out[1] = cv.y; out[2] = cv.z;
    part(L_TOES, out + 3, out + 6); part(R_TOES, out + 9, out + 12); part(TORSO, out + 15, out + 18); part(HEAD, out + 21, out + 24);
    part(L_LOWER_ARM, out + 27, out + 30); part(R_LOWER_ARM, out + 33, out + 36);
    std::memcpy(out + 39, env.smoothed, sizeof(env.smoothed));
    return out[21 + 1] < 1.f;
}

// training_locomotion::getBodyPartTarget / readPartDifference / getReward (learned_locomotion.cpp:248-345)
void partPoints(int e, int p, v3 pos[6], v3 vel[6], q4& localRot) {
    const uint32_t ent = entityOf(e, p);
    const v3 tp = entityPos(ent); const q4 tr = entityRot(ent);
    const v3 cog = globalCOG(ent, p), lv = entityLin(ent), av = entityAng(ent);
    for (int i = 0; i < 6; ++i) {
        v3 gp = tr * g.envs[e].localPositions[p][i] + tp;     // transformPosition (scale 1)
        pos[i] = gp;
        vel[i] = lv + cross(av, gp - cog);                   // getGlobalPointVelocity
    }
    q4 parentRot = kParent[p] >= 0 ? entityRot(entityOf(e, kParent[p])) : q4{0.f, 0.f, 0.f, 1.f};
    localRot = tr * conjugate(parentRot);
}
float rewardOf(int e) {
    const Env& env = g.envs[e];
    float positionError = 0.f, velocityError = 0.f, rotationError = 0.f;
    for (int p = 0; p < kParts; ++p) {
        v3 pos[6], vel[6]; q4 localRot;
        partPoints(e, p, pos, vel, localRot);
        float pe = 0.f, ve = 0.f;
        for (int i = 0; i < 6; ++i) { pe += length(pos[i] - env.targets[p].pos[i]); ve += length(vel[i] - env.targets[p].vel[i]); }
        q4 diff = env.targets[p].localRot * conjugate(localRot);
        positionError += pe; velocityError += ve;
        rotationError += 2.f * std::acos(clampf(diff.w, -1.f, 1.f));
    }
    float vcmError = length(entityLin(entityOf(e, TORSO)) - env.torsoVelocityTarget);
    float rp = std::exp(-10.f / kParts * positionError), rv = std::exp(-1.f / kParts * velocityError);
    float rlocal = std::exp(-10.f / kParts * rotationError), rvcm = std::exp(-vcmError);
    float headHeight = entityPos(entityOf(e, HEAD)).y;
    float fall = clampf(1.3f - 1.4f * (env.headTargetHeight - headHeight), 0.f, 1.f);
    return fall * (rp + rv + rlocal + rvcm);
}

// training_locomotion::reset + learned_locomotion::reset (learned_locomotion.cpp:294-307, 35-43) for the listed environments,
// which must be standing in their initial pose with refreshed transforms
void resetEnvState(int e) {
    Env& env = g.envs[e];
    for (int p = 0; p < kParts; ++p) {
        localPositionsOf(p, env.localPositions[p]);
        partPoints(e, p, env.targets[p].pos, env.targets[p].vel, env.targets[p].localRot);
    }
    std::memset(env.smoothed, 0, sizeof(env.smoothed));
    env.headTargetHeight = entityPos(entityOf(e, HEAD)).y;
    env.torsoVelocityTarget = {0.f, 0.f, 0.f};
    env.totalReward = 0.f;
}

bool buildWorld(int n) {
    destroyWorld();
    g.error.clear();
    if (!ok(physCreateWorld(g.device, &g.world), "world_create")) return false;
    g.n = n;
    g.envs.assign(n, Env{});
    const int side = (int)std::ceil(std::sqrt((double)n));
    const float spacing = 50.f;   // ground slabs are 40 m x 40 m
    // entities: per environment the ground (learned_locomotion.cpp:439-443) and the 14 body parts at the BASE pose
    std::vector<mi_entity_desc> ents((size_t)n * kEntitiesPerEnv);
    for (int e = 0; e < n; ++e) {
        Env& env = g.envs[e];
        env.origin = {(float)(e % side) * spacing, 0.f, (float)(e / side) * spacing};
        env.rng.state = g.seed + 0x632BE59BD9B4E019ull * (uint64_t)(e + 1);
        mi_entity_desc& gd = ents[(size_t)e * kEntitiesPerEnv];
        std::memset(&gd, 0, sizeof(gd));
        gd.position[0] = env.origin.x; gd.position[1] = -4.f; gd.position[2] = env.origin.z; gd.rotation[3] = 1.f; gd.kind = MI_ENTITY_STATIC;
        for (int p = 0; p < kParts; ++p) {
            mi_entity_desc& d = ents[(size_t)e * kEntitiesPerEnv + 1 + p];
            std::memset(&d, 0, sizeof(d));
            v3 bp = kScale * kPartDefs[p].pos; q4 br = axisAngle({0.f, 0.f, 1.f}, deg2rad(kPartDefs[p].zDeg));
            d.position[0] = bp.x; d.position[1] = bp.y; d.position[2] = bp.z;
            d.rotation[0] = br.x; d.rotation[1] = br.y; d.rotation[2] = br.z; d.rotation[3] = br.w;
            d.gravity_factor = 1.f; d.linear_damping = 0.4f; d.angular_damping = 0.4f;   // rigid_body_component defaults (rigid_body.h:20-21)
            d.kind = MI_ENTITY_DYNAMIC;
        }
    }
    uint32_t first = 0;
    if (!ok(PHYS(entities_create)(g.world, (uint32_t)ents.size(), ents.data(), &first), "entities_create")) return false;
    // colliders in creation order: ground, then the parts' colliders (material: flesh 0.2 / 1 / 985; ground metal 0.1 / 1 / 4)
    std::vector<uint32_t> colEnt; std::vector<mi_collider_desc> cols;
    for (int e = 0; e < n; ++e) {
        mi_collider_desc gc; std::memset(&gc, 0, sizeof(gc));
        gc.type = MI_COLLIDER_AABB; gc.restitution = 0.1f; gc.friction = 1.f; gc.density = 4.f;
        const float mn[3] = {-20.f, -4.f, -20.f}, mx[3] = {20.f, 4.f, 20.f};
        std::memcpy(gc.shape, mn, 12); std::memcpy(gc.shape + 3, mx, 12);
        colEnt.push_back((uint32_t)(e * kEntitiesPerEnv)); cols.push_back(gc);
        for (int c = 0; c < kNumCols; ++c) {
            const ColDef& d = kColDefs[c];
            mi_collider_desc cd; std::memset(&cd, 0, sizeof(cd));
            cd.restitution = 0.2f; cd.friction = 1.f; cd.density = 985.f;
            if (d.box) {
                cd.type = MI_COLLIDER_AABB;
                v3 ctr = kScale * d.a, r = kScale * d.b, lo = ctr - r, hi = ctr + r;
                const float s[6] = {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z}; std::memcpy(cd.shape, s, sizeof(s));
            } else {
                cd.type = MI_COLLIDER_CAPSULE;
                v3 a = kScale * d.a, b = kScale * d.b;
                const float s[7] = {a.x, a.y, a.z, b.x, b.y, b.z, kScale * d.r}; std::memcpy(cd.shape, s, sizeof(s));
            }
            colEnt.push_back(entityOf(e, d.part)); cols.push_back(cd);
        }
    }
    if (!ok(PHYS(colliders_add)(g.world, (uint32_t)cols.size(), colEnt.data(), cols.data()), "colliders_add")) return false;
    // joints from global points while every ragdoll still stands at the base pose (ragdoll.cpp:100-116)
    g.coneIds.assign((size_t)n * kCone, 0); g.hingeIds.assign((size_t)n * kHinge, 0);
    for (int e = 0; e < n; ++e)
        for (const JointDef& j : kJointDefs) {
            v3 ap = kScale * kPartDefs[j.anchorPart].pos; q4 ar = axisAngle({0.f, 0.f, 1.f}, deg2rad(kPartDefs[j.anchorPart].zDeg));
            v3 anchor = ar * (kScale * j.anchor) + ap;                                        // transformPosition(partTransform, scale * local)
            v3 axis = j.normalizeAxis ? normalize(j.axis) : j.axis;
            if (j.axisPart >= 0) axis = axisAngle({0.f, 0.f, 1.f}, deg2rad(kPartDefs[j.axisPart].zDeg)) * j.axis;   // transformDirection
            const float anchor3[3] = {anchor.x, anchor.y, anchor.z}, axis3[3] = {axis.x, axis.y, axis.z};
            const float l0 = j.l0 <= -1000.f ? -1.f : deg2rad(j.l0), l1 = deg2rad(j.l1);
            uint32_t id = 0;
            if (!ok(PHYS(constraint_create_from_global)(g.world, j.cone ? MI_CONSTRAINT_CONE_TWIST : MI_CONSTRAINT_HINGE, entityOf(e, j.a), entityOf(e, j.b), anchor3, axis3, l0, l1, &id),
                    "constraint_create_from_global")) return false;
            (j.cone ? g.coneIds[(size_t)e * kCone + j.slot] : g.hingeIds[(size_t)e * kHinge + j.slot]) = id;
        }
    g.cones.resize((size_t)n * kCone); g.hinges.resize((size_t)n * kHinge);
    for (size_t i = 0; i < g.cones.size(); ++i) if (!ok(PHYS(constraint_get)(g.world, MI_CONSTRAINT_CONE_TWIST, g.coneIds[i], &g.cones[i], sizeof(g.cones[i])), "constraint_get")) return false;
    for (size_t i = 0; i < g.hinges.size(); ++i) if (!ok(PHYS(constraint_get)(g.world, MI_CONSTRAINT_HINGE, g.hingeIds[i], &g.hinges[i], sizeof(g.hinges[i])), "constraint_get")) return false;
    // local centres of gravity (identical for every environment)
    {
        const uint32_t ne = (uint32_t)(n * kEntitiesPerEnv);
        std::vector<float> cog(3 * (size_t)ne);
        if (!ok(PHYS(world_get_mass_properties)(g.world, nullptr, nullptr, cog.data(), ne), "world_get_mass_properties")) return false;
        for (int p = 0; p < kParts; ++p) g.localCOG[p] = {cog[3 * entityOf(0, p)], cog[3 * entityOf(0, p) + 1], cog[3 * entityOf(0, p) + 2]};
    }
    // initial pose: humanoid_ragdoll::create(scene, vec3(0, 1.25, 0)) (rotation 0), shifted to the environment's origin
    g.initialStates.assign((size_t)n * kParts * MI_BODY_STATE_FLOATS, 0.f);
    const q4 yaw = axisAngle({0.f, 1.f, 0.f}, 0.f);
    for (int e = 0; e < n; ++e)
        for (int p = 0; p < kParts; ++p) {
            float* st = &g.initialStates[((size_t)e * kParts + p) * MI_BODY_STATE_FLOATS];
            q4 br = axisAngle({0.f, 0.f, 1.f}, deg2rad(kPartDefs[p].zDeg));
            q4 r = yaw * br;
            v3 pp = yaw * (kScale * kPartDefs[p].pos) + (v3{0.f, 1.25f, 0.f} + g.envs[e].origin);
            st[0] = pp.x; st[1] = pp.y; st[2] = pp.z; st[3] = r.x; st[4] = r.y; st[5] = r.z; st[6] = r.w;
        }
    return true;
}

// Puts the listed environments back to the initial pose and resets their episode state.
bool resetEnvs(const std::vector<int>& list) {
    if (list.empty()) return true;
    std::vector<uint32_t> ents; std::vector<float> states;
    for (int e : list)
        for (int p = 0; p < kParts; ++p) {
            ents.push_back(entityOf(e, p));
            const float* st = &g.initialStates[((size_t)e * kParts + p) * MI_BODY_STATE_FLOATS];
            states.insert(states.end(), st, st + MI_BODY_STATE_FLOATS);
        }
    if (!ok(PHYS(world_set_body_states)(g.world, (uint32_t)ents.size(), ents.data(), states.data()), "world_set_body_states")) return false;
    // transform_component of the reset parts = the new pose (the interpolated transforms only change in physicsStep)
    for (size_t i = 0; i < ents.size(); ++i) {
        const uint32_t e = ents[i]; const float* st = &states[i * MI_BODY_STATE_FLOATS];
        for (int k = 0; k < 3; ++k) { g.pos[3 * e + k] = st[k]; g.lin[3 * e + k] = 0.f; g.ang[3 * e + k] = 0.f; }
        for (int k = 0; k < 4; ++k) g.rot[4 * e + k] = st[3 + k];
    }
    for (int e : list) resetEnvState(e);
    return applyActions(nullptr, &list);   // learned_locomotion::reset: applyAction({}) arms the position motors
}

bool ensureBatch(int n) {
    if (g.world && g.n == n) return true;
    if (!buildWorld(n)) return false;
    if (!refreshTransforms()) return false;
    std::vector<int> all(n); for (int e = 0; e < n; ++e) all[e] = e;
    return resetEnvs(all);
}

// MI_LEARN_PROFILE=1: wall time per phase of the batched step, printed every 100 steps (development aid)
struct PhaseTimer {
    double acc[6] = {0, 0, 0, 0, 0, 0}; int steps = 0; bool on = std::getenv("MI_LEARN_PROFILE") != nullptr;
    std::chrono::steady_clock::time_point t0;
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void lap(int k) { if (!on) return; auto t = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double, std::milli>(t - t0).count(); t0 = t; }
    void end() {
        if (!on || ++steps % 100) return;
        std::fprintf(stderr, "[learning] per step (ms): actions %.3f pushes %.3f physics %.3f readback %.3f state+reward %.3f resets %.3f\n",
                     acc[0] / 100, acc[1] / 100, acc[2] / 100, acc[3] / 100, acc[4] / 100, acc[5] / 100);
        for (double& a : acc) a = 0;
    }
} g_timer;

// updatePhysics (learned_locomotion.cpp:452-489) for every environment
bool stepAll(const float* actions, float* outStates, float* outRewards, int* outDone) {
    g_timer.start();
    if (!applyActions(actions)) return false;
    g_timer.lap(0);
    // random pushes: with probability 0.02 a ray from 5 m away at a random body part (458-468), strength 1000
    std::vector<float> origins, directions; std::vector<uint32_t> ranges;
    for (int e = 0; e < g.n; ++e) {
        Rng& rng = g.envs[e].rng;
        if (rng.float01() < 0.02f) {
            uint32_t part = rng.between(0, kParts - 1);
            v3 target = entityPos(entityOf(e, (int)part)) + v3{0.f, 0.2f, 0.f};
            float dx = rng.floatBetween(-1.f, 1.f), dz = rng.floatBetween(-1.f, 1.f);
            v3 dir = normalize(v3{dx, 0.f, dz});
            v3 origin = target - dir * 5.f;
            origins.insert(origins.end(), {origin.x, origin.y, origin.z}); directions.insert(directions.end(), {dir.x, dir.y, dir.z});
            ranges.push_back((uint32_t)(e * kEntitiesPerEnv)); ranges.push_back((uint32_t)((e + 1) * kEntitiesPerEnv));
        }
    }
    if (!origins.empty() && !ok(PHYS(world_test_interactions)(g.world, (uint32_t)(origins.size() / 3), origins.data(), directions.data(), nullptr, ranges.data()), "world_test_interactions")) return false;
    // physicsStep(scene, arena, timer = 0, settings{frameRate 60}, 1/60): one internal step; the interpolated transforms end
    // up at physics_transform0, i.e. the pose BEFORE this step, while the velocities are the new ones (physics.cpp:1364-1402)
    mi_step_settings settings; std::memset(&settings, 0, sizeof(settings));
    settings.fixed_frame_rate = 1; settings.frame_rate = 60; settings.max_physics_iterations_per_frame = 4; settings.num_rigid_solver_iterations = 30;
    g_timer.lap(1);
    if (!ok(PHYS(world_step)(g.world, &settings, 1.f / 60.f), "world_step")) return false;
    g_timer.lap(2);
    if (!refreshTransforms()) return false;
    g_timer.lap(3);
    std::vector<int> failed;
    std::vector<uint8_t> fell(g.n, 0);
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g.n >= 256)   // environments are independent: the host-side state / reward arithmetic scales over the cores
    for (int e = 0; e < g.n; ++e) {
        float state[kStateFloats];
        bool failure = stateOf(e, state);
        float reward = 0.f;
        if (!failure) { reward = rewardOf(e); g.envs[e].totalReward += reward; }
        else fell[e] = 1;
        if (outStates) std::memcpy(outStates + (size_t)e * kStateFloats, state, sizeof(state));
        if (outRewards) outRewards[e] = reward;
        if (outDone) outDone[e] = failure ? 1 : 0;
    }
    for (int e = 0; e < g.n; ++e) if (fell[e]) failed.push_back(e);
    g_timer.lap(4);
    const bool done = outDone ? resetEnvs(failed) : true;   // the batch API resets fallen ragdolls itself; the single-environment API leaves that to resetPhysics
    g_timer.lap(5); g_timer.end();
    return done;
}

}  // namespace

EXPORT int getPhysicsStateSize() { return kStateFloats; }
EXPORT int getPhysicsActionSize() { return kActionFloats; }

// learned_locomotion.cpp:398-427: no limits for the state; action limits from the constraints' limits
EXPORT void getPhysicsRanges(float* stateMin, float* stateMax, float* actionMin, float* actionMax) {
    for (int i = 0; i < kStateFloats; ++i) { stateMin[i] = -FLT_MAX; stateMax[i] = FLT_MAX; }
    if (!g.world && !ensureBatch(1)) { for (int i = 0; i < kActionFloats; ++i) { actionMin[i] = -kPi; actionMax[i] = kPi; } return; }
    int k = 0;
    for (int s = 0; s < kCone; ++s) {
        const mi_cone_twist_constraint& c = g.cones[s];
        actionMin[k] = c.twist_limit >= 0.f ? -c.twist_limit : -kPi; actionMax[k++] = c.twist_limit >= 0.f ? c.twist_limit : kPi;
        actionMin[k] = c.swing_limit >= 0.f ? -c.swing_limit : -kPi; actionMax[k++] = c.swing_limit >= 0.f ? c.swing_limit : kPi;
        actionMin[k] = -kPi; actionMax[k++] = kPi;
    }
    for (int s = 0; s < kHinge; ++s) {
        const mi_hinge_constraint& h = g.hinges[s];
        actionMin[k] = h.min_rotation_limit <= 0.f ? h.min_rotation_limit : -kPi; actionMax[k++] = h.max_rotation_limit >= 0.f ? h.max_rotation_limit : kPi;
    }
}

EXPORT void resetPhysics(float* outState) {
    if (!ensureBatch(1)) return;
    std::vector<int> all{0};
    if (!resetEnvs(all)) return;
    if (outState) stateOf(0, outState);
}

EXPORT int updatePhysics(float* action, float* outState, float* outReward) {
    if (!g.world && !ensureBatch(1)) return 1;
    int done = 0;
    float reward = 0.f;
    std::vector<float> states((size_t)g.n * kStateFloats), rewards(g.n);
    std::vector<float> actions((size_t)g.n * kActionFloats, 0.f);
    std::memcpy(actions.data(), action, sizeof(float) * kActionFloats);
    if (!stepAll(actions.data(), states.data(), rewards.data(), nullptr)) return 1;
    std::memcpy(outState, states.data(), sizeof(float) * kStateFloats);
    reward = rewards[0]; done = states[22] < 1.f ? 1 : 0;
    *outReward = reward;
    return done;
}

// ---- batched environments -----------------------------------------------------------------------------------------------
EXPORT void setPhysicsSeed(unsigned long long seed) { g.seed = seed; for (int e = 0; e < g.n; ++e) g.envs[e].rng.state = g.seed + 0x632BE59BD9B4E019ull * (uint64_t)(e + 1); }
EXPORT void setPhysicsDevice(int device) { if (device != g.device) { destroyWorld(); g.device = device; } }
EXPORT const char* getPhysicsError() { return g.error.c_str(); }
EXPORT int getPhysicsNumEnvs() { return g.n; }
// (Re)creates `numEnvs` environments and writes their initial states ([numEnvs][stateSize]); 0 on success.
EXPORT int resetPhysicsBatch(int numEnvs, float* outStates) {
    if (numEnvs <= 0) return MI_ERR_INVALID_ARGUMENT;
    if (!ensureBatch(numEnvs)) return MI_ERR_DEVICE;
    std::vector<int> all(numEnvs); for (int e = 0; e < numEnvs; ++e) all[e] = e;
    if (!resetEnvs(all)) return MI_ERR_DEVICE;
    if (outStates) for (int e = 0; e < numEnvs; ++e) stateOf(e, outStates + (size_t)e * kStateFloats);
    return MI_OK;
}
// One updatePhysics for every environment: actions [numEnvs][actionSize] -> states, rewards, done flags.  An environment whose
// ragdoll fell (done = 1) is reset in place after its terminal state was written; 0 on success.
EXPORT int updatePhysicsBatch(const float* actions, float* outStates, float* outRewards, int* outDone) {
    if (!g.world) return MI_ERR_INVALID_ARGUMENT;
    std::vector<int> done(g.n);
    if (!stepAll(actions, outStates, outRewards, outDone ? outDone : done.data())) return MI_ERR_DEVICE;
    return MI_OK;
}
EXPORT void shutdownPhysics() { destroyWorld(); }
