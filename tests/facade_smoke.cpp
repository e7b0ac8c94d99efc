// Compile-and-link check of include/physics_world.hpp against libmi_physics.so (run by tests/test_capi_symbols.py;
// executing it needs a GPU, compiling/linking does not).
#include <cstdio>
#include <cstdlib>
#include "physics_world.hpp"
using namespace mi_facade;
int main() {
    try {
        physics_world world(0);
        physics_material mat{0.1f, 0.5f, 1.f};
        auto ground = world.addStaticCollider(trs{}, {collider_component::asAABB({-50, -4, -50}, {50, 0, 50}, mat)});
        (void)ground;
        trs t; t.position = {0, 2, 0};
        auto a = world.addRigidBody(t, rigid_body_component{}, {collider_component::asSphere({0, 0, 0}, 0.5f, mat)});
        t.position = {1.2f, 2, 0};
        auto b = world.addRigidBody(t, rigid_body_component{}, {collider_component::asAABB({-0.5f, -0.5f, -0.5f}, {0.5f, 0.5f, 0.5f}, mat)});
        auto h = world.addHingeConstraintFromGlobalPoints(a, b, {0.6f, 2, 0}, {0, 0, 1}, -0.5f, 0.5f);
        auto pod = world.getConstraint<mi_hinge_constraint>(h);
        pod.max_motor_torque = 5.f; pod.motor_velocity_or_target_angle = 1.f;
        world.setConstraint(h, pod);
        physics_settings settings;
        int begins = 0, ends = 0;
        settings.collisionBeginCallback = [&](const collision_begin_event& e) { ++begins; (void)e; };
        settings.collisionEndCallback = [&](const collision_end_event& e) { ++ends; (void)e; };
        // the wider surface: trigger + force field entities, a cloth, a ray push, constraint deletion
        int enters = 0;
        trs tt; tt.position = {0, 1, 0};
        world.addTrigger(tt, {collider_component::asSphere({0, 0, 0}, 1.5f, mat)}, [&](const trigger_event& e) { if (e.type == trigger_event_enter) ++enters; });
        world.addForceField(trs{}, {0.5f, 0.f, 0.f});
        uint32_t cloth = world.addCloth(2.f, 2.f, 8, 8, 1.f);
        trs ct; ct.position = {5, 4, 0};
        world.setWorldPositionOfFixedVertices(cloth, ct, true);
        for (int i = 0; i < 60; ++i) physicsStep(world, settings, 1.f / 60.f);
        world.testPhysicsInteraction(ray{{-5.f, 0.5f, 0.f}, {1.f, 0.f, 0.f}}, 200.f);
        world.deleteConstraint(h);
        for (int i = 0; i < 60; ++i) physicsStep(world, settings, 1.f / 60.f);
        auto tr = world.transforms();
        auto cp = world.clothPositions(cloth, 64);
        if (begins < 1) { std::printf("facade error: no collision-begin callback fired\n"); return 1; }
        if (enters < 1) { std::printf("facade error: no trigger-enter callback fired\n"); return 1; }
        if (!(cp[63].y < 3.9f)) { std::printf("facade error: the cloth did not move\n"); return 1; }
        if (std::getenv("MI_FACADE_SHARD")) {   // the multi-GPU surface with ONE rank (opt-in: the default run of this check stays what it was)
            physics_world tile(0);
            tile.addStaticCollider(trs{}, {collider_component::asAABB({-50, -4, -50}, {50, 0, 50}, mat)});
            trs tt; tt.position = {0, 1, 0};
            tile.addRigidBody(tt, rigid_body_component{}, {collider_component::asSphere({0, 0, 0}, 0.5f, mat)});
            mi_shard_desc d{}; d.rank = 0; d.num_ranks = 1; d.tiles_x = d.tiles_z = 1; d.origin_x = d.origin_z = -10.f; d.tile_size_x = d.tile_size_z = 20.f; d.ghost_margin = 2.f;
            tile.enableSharding(d);
            tile.attachRccl(physics_world::shardUniqueId());
            for (int i = 0; i < 30; ++i) tile.stepFixed(settings, 1.f / 120.f);
            tile.rebalance();
            tile.setExactSeam(true);                        // per-sweep hand-over through the library transport (no neighbour here: nothing to send)
            for (int i = 0; i < 5; ++i) tile.stepFixed(settings, 1.f / 120.f);
            tile.setExactSeam(false);
            auto g = tile.globalCounts();
            if (tile.ownedEntities().size() != 1 || g[0] != 1) { std::printf("facade error: sharded world with one rank\n"); return 1; }
        }
        std::printf("facade ok: a.y=%f b.y=%f contacts=%u begins=%d ends=%d\n", tr[a.id].position.y, tr[b.id].position.y, world.counts().num_contacts, begins, ends);
    } catch (const std::exception& e) {
        std::printf("facade error: %s\n", e.what());
        return 1;
    }
    return 0;
}
