// physics_world.hpp — header-only C++ facade over the C ABI (mi_physics.h) carrying the names BASELINE.json's
// north_star uses (physics_world::step / addRigidBody / addConstraint) and the reference's own free-function and
// component names (physicsStep, physics_settings, rigid_body_component, collider_component::asXxx,
// addXxxConstraintFromGlobalPoints — src/physics/physics.h:108-264, 382-405; src/physics/rigid_body.h:18-46).
// The reference stores components in an EnTT registry; EnTT is not vendored, so entities here are plain handles
// owned by the world.  INTEGRATION.md shows how the reference's scene hooks would forward to these calls.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include <functional>
#include <array>
#include "mi_physics.h"
#include "mi_constraints.h"
#include "mi_shard.h"

namespace mi_facade {

struct vec3 { float x = 0, y = 0, z = 0; };
struct quat { float x = 0, y = 0, z = 0, w = 1; };
struct trs { vec3 position; quat rotation; };                       // transform_component (scale is unused by physics)

struct physics_material { float restitution = 0.1f, friction = 0.5f, density = 1.f; };   // src/physics/physics.h:40-47

// physics_settings (src/physics/physics.h:382-400) minus the std::function callbacks.
// collision_begin_event / collision_end_event — src/physics/physics.h:356-380 (entities and colliders by id)
struct collision_begin_event { uint32_t entityA, entityB, colliderA, colliderB; vec3 position, normal, relativeVelocity; };
struct collision_end_event { uint32_t entityA, entityB, colliderA, colliderB; };

enum trigger_event_type { trigger_event_enter, trigger_event_leave };                    // src/physics/physics.h:187-198
struct trigger_event { uint32_t trigger, other; trigger_event_type type; };
struct ray { vec3 origin, direction; };

struct physics_settings {
    bool fixedFrameRate = true;
    uint32_t frameRate = 120;
    uint32_t maxPhysicsIterationsPerFrame = 4;
    uint32_t numRigidSolverIterations = 30;
    std::function<void(const collision_begin_event&)> collisionBeginCallback;   // src/physics/physics.h:398-399
    std::function<void(const collision_end_event&)> collisionEndCallback;
    mi_step_settings c() const { return mi_step_settings{fixedFrameRate ? 1u : 0u, frameRate, maxPhysicsIterationsPerFrame, numRigidSolverIterations}; }
};

// rigid_body_component(bool kinematic, float gravityFactor, float linearDamping, float angularDamping) — rigid_body.h:20-21
struct rigid_body_component {
    bool kinematic = false;
    float gravityFactor = 1.f, linearDamping = 0.4f, angularDamping = 0.4f;
    vec3 linearVelocity, angularVelocity;
};

// collider_component::asSphere / asCapsule / asCylinder / asAABB / asOBB / asHull — physics.h:110-157
struct collider_component {
    mi_collider_desc d{};
    static collider_component make(uint32_t type, physics_material m) {
        collider_component c; c.d.type = type; c.d.restitution = m.restitution; c.d.friction = m.friction; c.d.density = m.density; return c;
    }
    static collider_component asSphere(vec3 center, float radius, physics_material m) {
        auto c = make(MI_COLLIDER_SPHERE, m); float s[4] = {center.x, center.y, center.z, radius}; for (int i = 0; i < 4; ++i) c.d.shape[i] = s[i]; return c;
    }
    static collider_component asCapsule(vec3 a, vec3 b, float radius, physics_material m) {
        auto c = make(MI_COLLIDER_CAPSULE, m); float s[7] = {a.x, a.y, a.z, b.x, b.y, b.z, radius}; for (int i = 0; i < 7; ++i) c.d.shape[i] = s[i]; return c;
    }
    static collider_component asCylinder(vec3 a, vec3 b, float radius, physics_material m) {
        auto c = asCapsule(a, b, radius, m); c.d.type = MI_COLLIDER_CYLINDER; return c;
    }
    static collider_component asAABB(vec3 minCorner, vec3 maxCorner, physics_material m) {
        auto c = make(MI_COLLIDER_AABB, m); float s[6] = {minCorner.x, minCorner.y, minCorner.z, maxCorner.x, maxCorner.y, maxCorner.z};
        for (int i = 0; i < 6; ++i) c.d.shape[i] = s[i]; return c;
    }
    static collider_component asOBB(quat rotation, vec3 center, vec3 radius, physics_material m) {
        auto c = make(MI_COLLIDER_OBB, m);
        float s[10] = {rotation.x, rotation.y, rotation.z, rotation.w, center.x, center.y, center.z, radius.x, radius.y, radius.z};
        for (int i = 0; i < 10; ++i) c.d.shape[i] = s[i]; return c;
    }
    static collider_component asHull(quat rotation, vec3 position, uint32_t geometryIndex, physics_material m) {
        auto c = make(MI_COLLIDER_HULL, m); float s[7] = {rotation.x, rotation.y, rotation.z, rotation.w, position.x, position.y, position.z};
        for (int i = 0; i < 7; ++i) c.d.shape[i] = s[i]; c.d.hull_geometry = geometryIndex; return c;
    }
};

struct scene_entity { uint32_t id = 0xFFFFFFFFu; };
struct constraint_handle { uint32_t type, id; };

class physics_world {
public:
    explicit physics_world(int device = 0) {
        mi_world_desc d{device, 0};
        check(mi_world_create(&d, &w_), "mi_world_create");
    }
    ~physics_world() { if (w_) mi_world_destroy(w_); }
    physics_world(const physics_world&) = delete;
    physics_world& operator=(const physics_world&) = delete;

    // createEntity().addComponent<transform_component>(t).addComponent<collider_component>(...)...addComponent<rigid_body_component>(rb)
    scene_entity addRigidBody(const trs& t, const rigid_body_component& rb, const std::vector<collider_component>& colliders) {
        return addEntity(t, rb.kinematic ? MI_ENTITY_KINEMATIC : MI_ENTITY_DYNAMIC, &rb, colliders);
    }
    // An entity with colliders but no rigid_body_component: static colliders (dummy body, physics.cpp:1214).
    scene_entity addStaticCollider(const trs& t, const std::vector<collider_component>& colliders) { return addEntity(t, MI_ENTITY_STATIC, nullptr, colliders); }
    uint32_t allocateBoundingHullGeometry(const std::vector<vec3>& vertices, const std::vector<std::array<uint32_t, 3>>& triangles) {
        uint32_t g = 0;
        check(mi_hull_geometry_create(w_, &vertices[0].x, (uint32_t)vertices.size(), triangles.empty() ? nullptr : &triangles[0][0], (uint32_t)triangles.size(), &g),
              "mi_hull_geometry_create");
        return g;
    }

    // addConstraint(a, b, const xxx_constraint&) — physics.h:235-240
    constraint_handle addConstraint(scene_entity a, scene_entity b, const mi_distance_constraint& c) { return add(MI_CONSTRAINT_DISTANCE, a, b, &c, sizeof(c)); }
    constraint_handle addConstraint(scene_entity a, scene_entity b, const mi_ball_constraint& c) { return add(MI_CONSTRAINT_BALL, a, b, &c, sizeof(c)); }
    constraint_handle addConstraint(scene_entity a, scene_entity b, const mi_fixed_constraint& c) { return add(MI_CONSTRAINT_FIXED, a, b, &c, sizeof(c)); }
    constraint_handle addConstraint(scene_entity a, scene_entity b, const mi_hinge_constraint& c) { return add(MI_CONSTRAINT_HINGE, a, b, &c, sizeof(c)); }
    constraint_handle addConstraint(scene_entity a, scene_entity b, const mi_cone_twist_constraint& c) { return add(MI_CONSTRAINT_CONE_TWIST, a, b, &c, sizeof(c)); }
    constraint_handle addConstraint(scene_entity a, scene_entity b, const mi_slider_constraint& c) { return add(MI_CONSTRAINT_SLIDER, a, b, &c, sizeof(c)); }

    // add*ConstraintFromGlobalPoints — physics.h:217-233
    constraint_handle addDistanceConstraintFromGlobalPoints(scene_entity a, scene_entity b, vec3 globalAnchorA, vec3 globalAnchorB) {
        return fromGlobal(MI_CONSTRAINT_DISTANCE, a, b, globalAnchorA, &globalAnchorB, 0.f, 0.f);
    }
    constraint_handle addBallConstraintFromGlobalPoints(scene_entity a, scene_entity b, vec3 globalAnchor) { return fromGlobal(MI_CONSTRAINT_BALL, a, b, globalAnchor, nullptr, 0.f, 0.f); }
    constraint_handle addFixedConstraintFromGlobalPoints(scene_entity a, scene_entity b, vec3 globalAnchor) { return fromGlobal(MI_CONSTRAINT_FIXED, a, b, globalAnchor, nullptr, 0.f, 0.f); }
    constraint_handle addHingeConstraintFromGlobalPoints(scene_entity a, scene_entity b, vec3 globalAnchor, vec3 globalHingeAxis, float minLimit = 1.f, float maxLimit = -1.f) {
        return fromGlobal(MI_CONSTRAINT_HINGE, a, b, globalAnchor, &globalHingeAxis, minLimit, maxLimit);
    }
    constraint_handle addConeTwistConstraintFromGlobalPoints(scene_entity a, scene_entity b, vec3 globalAnchor, vec3 globalAxis, float swingLimit, float twistLimit) {
        return fromGlobal(MI_CONSTRAINT_CONE_TWIST, a, b, globalAnchor, &globalAxis, swingLimit, twistLimit);
    }
    constraint_handle addSliderConstraintFromGlobalPoints(scene_entity a, scene_entity b, vec3 globalAnchor, vec3 globalAxis, float minLimit = 1.f, float maxLimit = -1.f) {
        return fromGlobal(MI_CONSTRAINT_SLIDER, a, b, globalAnchor, &globalAxis, minLimit, maxLimit);
    }
    // getConstraint(scene, handle) returns a mutable reference in the reference; here: read, edit, write back.
    template <class Pod> Pod getConstraint(constraint_handle h) { Pod p; check(mi_constraint_get(w_, h.type, h.id, &p, sizeof(p)), "mi_constraint_get"); return p; }
    template <class Pod> void setConstraint(constraint_handle h, const Pod& p) { check(mi_constraint_update(w_, h.type, h.id, &p, sizeof(p)), "mi_constraint_update"); }

    void applyForce(scene_entity e, vec3 force, vec3 torque) { check(mi_entity_apply_force(w_, e.id, &force.x, &torque.x), "mi_entity_apply_force"); }
    // testPhysicsInteraction(scene, ray, strength) — src/physics/physics.h:404
    void testPhysicsInteraction(ray r, float strength = 1000.f) { check(mi_world_test_interactions(w_, 1, &r.origin.x, &r.direction.x, &strength, nullptr), "mi_world_test_interactions"); }

    // deleteConstraint / deleteAllConstraintsFromEntity / deleteAllConstraints — src/physics/physics.h:251-260
    void deleteConstraint(constraint_handle h) { check(mi_constraint_destroy(w_, h.type, h.id), "mi_constraint_destroy"); }
    void deleteAllConstraintsFromEntity(scene_entity e) { check(mi_entity_destroy_constraints(w_, e.id), "mi_entity_destroy_constraints"); }
    void deleteAllConstraints() { check(mi_constraints_destroy_all(w_), "mi_constraints_destroy_all"); }

    // force_field_component / trigger_component entities (src/physics/physics.h:182-203): colliders define the volume; a force
    // field without colliders is global.  The trigger callback fires from step() like the collision callbacks.
    scene_entity addForceField(const trs& t, vec3 force, const std::vector<collider_component>& colliders = {}) {
        scene_entity e = addEntity(t, MI_ENTITY_FORCE_FIELD, nullptr, colliders);
        check(mi_entity_set_force(w_, e.id, &force.x), "mi_entity_set_force");
        return e;
    }
    scene_entity addTrigger(const trs& t, const std::vector<collider_component>& colliders, std::function<void(const trigger_event&)> callback) {
        scene_entity e = addEntity(t, MI_ENTITY_TRIGGER, nullptr, colliders);
        triggerCallbacks_.emplace_back(e.id, std::move(callback));
        return e;
    }

    // heightmap_collider_component (src/terrain/heightmap_collider.h:126-151): one per world; heights = 129 x 129 uint16 per chunk
    void addHeightmap(uint32_t chunksPerDim, float chunkSize, physics_material m) { check(mi_heightmap_create(w_, chunksPerDim, chunkSize, m.restitution, m.friction), "mi_heightmap_create"); }
    void setHeightmapChunk(uint32_t x, uint32_t z, const uint16_t* heights) { check(mi_heightmap_set_chunk_heights(w_, x, z, heights), "mi_heightmap_set_chunk_heights"); }
    void updateHeightmap(vec3 minCorner, float amplitudeScale) { check(mi_heightmap_update(w_, &minCorner.x, amplitudeScale), "mi_heightmap_update"); }
    float getHeightAt(float x, float z) { float h = 0.f; check(mi_heightmap_get_height(w_, x, z, &h), "mi_heightmap_get_height"); return h; }

    // cloth_component (src/physics/cloth.h:5-60)
    uint32_t addCloth(float width, float height, uint32_t gridSizeX, uint32_t gridSizeY, float totalMass, float stiffness = 0.5f, float damping = 0.3f, float gravityFactor = 1.f) {
        mi_cloth_desc d{width, height, gridSizeX, gridSizeY, totalMass, stiffness, damping, gravityFactor};
        uint32_t id = 0; check(mi_cloth_create(w_, &d, &id), "mi_cloth_create"); return id;
    }
    void setWorldPositionOfFixedVertices(uint32_t cloth, const trs& t, bool moveRigid = false) {
        check(mi_cloth_set_fixed_vertices(w_, cloth, &t.position.x, &t.rotation.x, moveRigid ? 1u : 0u), "mi_cloth_set_fixed_vertices");
    }
    std::vector<vec3> clothPositions(uint32_t cloth, uint32_t numParticles) {
        std::vector<float> p(3 * (size_t)numParticles);
        check(mi_cloth_get_state(w_, cloth, p.data(), nullptr, numParticles), "mi_cloth_get_state");
        std::vector<vec3> out(numParticles);
        for (uint32_t i = 0; i < numParticles; ++i) out[i] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
        return out;
    }

    // physicsStep(scene, arena, timer, settings, dt) — the timer and the arena live inside the world.
    void step(const physics_settings& settings, float dt) { mi_step_settings s = settings.c(); syncEvents(settings); check(mi_world_step(w_, &s, dt), "mi_world_step"); fireEvents(settings); }
    void stepFixed(const physics_settings& settings, float dt, uint32_t n = 1) {
        mi_step_settings s = settings.c(); syncEvents(settings); check(mi_world_step_fixed(w_, &s, dt, n), "mi_world_step_fixed"); fireEvents(settings);
    }

    uint32_t numEntities() { uint32_t n = 0; check(mi_world_num_entities(w_, &n), "mi_world_num_entities"); return n; }
    std::vector<trs> transforms() {
        uint32_t n = numEntities();
        std::vector<float> p(3 * n), q(4 * n);
        check(mi_world_get_transforms(w_, p.data(), q.data(), n), "mi_world_get_transforms");
        std::vector<trs> out(n);
        for (uint32_t i = 0; i < n; ++i) { out[i].position = {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; out[i].rotation = {q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]}; }
        return out;
    }
    mi_step_counts counts() { mi_step_counts c; check(mi_world_get_counts(w_, &c), "mi_world_get_counts"); return c; }
    mi_world* handle() { return w_; }

    // One scene on several GPUs (include/mi_shard.h; the reference has no such path): every process builds the same scene and simulates one tile.
    void enableSharding(const mi_shard_desc& d) { check(mi_world_shard_enable(w_, &d), "mi_world_shard_enable"); }
    static std::array<char, 128> shardUniqueId() { std::array<char, 128> id{}; if (mi_shard_get_unique_id(id.data()) != MI_OK) throw std::runtime_error(std::string("mi_shard_get_unique_id: ") + mi_last_error()); return id; }
    void attachRccl(const std::array<char, 128>& id) { check(mi_world_shard_attach_rccl(w_, id.data()), "mi_world_shard_attach_rccl"); }   // neighbour exchange inside step()
    std::vector<scene_entity> ownedEntities() {            // whose read-backs are authoritative on this rank
        uint32_t n = 0; check(mi_world_shard_owned_entities(w_, nullptr, 0, &n), "mi_world_shard_owned_entities");
        std::vector<uint32_t> ids(n); if (n) check(mi_world_shard_owned_entities(w_, ids.data(), n, &n), "mi_world_shard_owned_entities");
        std::vector<scene_entity> out(n); for (uint32_t i = 0; i < n; ++i) out[i].id = ids[i];
        return out;
    }
    void rebalance(uint32_t bins = 256) { check(mi_world_shard_rebalance(w_, bins), "mi_world_shard_rebalance"); }   // load balance: every rank, between the same two steps
    // exact seam (option, slabs only): per-sweep hand-over inside step() with the library transport; a single world told the tiling gives the ranks' result bit for bit
    void setExactSeam(bool on, mi_shard_sweep_fn exchange = nullptr, void* user = nullptr) { check(mi_world_shard_set_exact_seam(w_, on ? 1u : 0u, exchange, user), "mi_world_shard_set_exact_seam"); }
    void setSeamTiling(const mi_shard_desc* d) { check(mi_world_set_seam_tiling(w_, d), "mi_world_set_seam_tiling"); }
    std::array<uint64_t, 3> globalCounts() {               // bodies, manifolds, contacts of the whole scene (one all-reduce)
        uint32_t b = 0, m = 0, c = 0; check(mi_world_shard_counts(w_, &b, &m, &c), "mi_world_shard_counts");
        std::array<uint64_t, 3> v{b, m, c}; check(mi_world_shard_allreduce_u64(w_, v.data(), 3), "mi_world_shard_allreduce_u64"); return v;
    }

private:
    // The reference fires its std::function callbacks inside the step (handleCollisionCallbacks, physics.cpp:1041-1178); here
    // the events of the step are polled right after it and delivered in the same order.
    void syncEvents(const physics_settings& settings) {
        bool want = (bool)settings.collisionBeginCallback || (bool)settings.collisionEndCallback || !triggerCallbacks_.empty();
        if (want != eventsOn_) { check(mi_world_enable_events(w_, want ? 1u : 0u), "mi_world_enable_events"); eventsOn_ = want; }
    }
    void fireEvents(const physics_settings& settings) {
        if (!eventsOn_) return;
        uint32_t n = 0;
        check(mi_world_poll_events(w_, nullptr, 0, &n), "mi_world_poll_events");
        if (!n) return;
        std::vector<mi_event> ev(n);
        check(mi_world_poll_events(w_, ev.data(), n, &n), "mi_world_poll_events");
        for (const mi_event& e : ev) {
            if (e.type == MI_EVENT_TRIGGER_ENTER || e.type == MI_EVENT_TRIGGER_LEAVE) {   // trigger_component::callback (physics.cpp:1004-1036)
                for (auto& tc : triggerCallbacks_)
                    if (tc.first == e.entity_a && tc.second) tc.second(trigger_event{e.entity_a, e.entity_b, e.type == MI_EVENT_TRIGGER_ENTER ? trigger_event_enter : trigger_event_leave});
            } else if (e.type == MI_EVENT_COLLISION_BEGIN) {
                if (settings.collisionBeginCallback)
                    settings.collisionBeginCallback(collision_begin_event{e.entity_a, e.entity_b, e.collider_a, e.collider_b, {e.point[0], e.point[1], e.point[2]},
                                                                          {e.normal[0], e.normal[1], e.normal[2]},
                                                                          {e.relative_velocity[0], e.relative_velocity[1], e.relative_velocity[2]}});
            } else if (settings.collisionEndCallback) settings.collisionEndCallback(collision_end_event{e.entity_a, e.entity_b, e.collider_a, e.collider_b});
        }
    }
    bool eventsOn_ = false;
    std::vector<std::pair<uint32_t, std::function<void(const trigger_event&)>>> triggerCallbacks_;
public:

private:
    mi_world* w_ = nullptr;
    static void check(int rc, const char* what) { if (rc != MI_OK) throw std::runtime_error(std::string(what) + ": " + mi_last_error()); }
    scene_entity addEntity(const trs& t, uint32_t kind, const rigid_body_component* rb, const std::vector<collider_component>& colliders) {
        mi_entity_desc d{};
        d.position[0] = t.position.x; d.position[1] = t.position.y; d.position[2] = t.position.z;
        d.rotation[0] = t.rotation.x; d.rotation[1] = t.rotation.y; d.rotation[2] = t.rotation.z; d.rotation[3] = t.rotation.w;
        d.gravity_factor = rb ? rb->gravityFactor : 1.f; d.linear_damping = rb ? rb->linearDamping : 0.4f; d.angular_damping = rb ? rb->angularDamping : 0.4f;
        if (rb) { d.linear_velocity[0] = rb->linearVelocity.x; d.linear_velocity[1] = rb->linearVelocity.y; d.linear_velocity[2] = rb->linearVelocity.z;
                  d.angular_velocity[0] = rb->angularVelocity.x; d.angular_velocity[1] = rb->angularVelocity.y; d.angular_velocity[2] = rb->angularVelocity.z; }
        d.kind = kind;
        scene_entity e;
        check(mi_entity_create(w_, &d, &e.id), "mi_entity_create");
        for (const collider_component& c : colliders) check(mi_collider_add(w_, e.id, &c.d, nullptr), "mi_collider_add");
        return e;
    }
    constraint_handle add(uint32_t type, scene_entity a, scene_entity b, const void* pod, uint32_t bytes) {
        constraint_handle h{type, 0};
        check(mi_constraint_create(w_, type, a.id, b.id, pod, bytes, &h.id), "mi_constraint_create");
        return h;
    }
    constraint_handle fromGlobal(uint32_t type, scene_entity a, scene_entity b, vec3 anchor, const vec3* axis, float l0, float l1) {
        constraint_handle h{type, 0};
        check(mi_constraint_create_from_global(w_, type, a.id, b.id, &anchor.x, axis ? &axis->x : nullptr, l0, l1, &h.id), "mi_constraint_create_from_global");
        return h;
    }
};

// Reference-named free function: physicsStep(game_scene&, memory_arena&, float& timer, const physics_settings&, float dt)
// (src/physics/physics.h:405).  The world owns what the scene, the arena and the timer were.
inline void physicsStep(physics_world& world, const physics_settings& settings, float dt) { world.step(settings, dt); }

}  // namespace mi_facade
