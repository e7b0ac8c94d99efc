// physics_mi355x.h — declarations of the backend stub (physics_mi355x.cpp); what src/physics/physics.h gains on the reference side.
#pragma once
struct game_scene; struct memory_arena; struct physics_settings; struct mi_world;
namespace entt { class registry; }

// drop-in for physicsStep (src/physics/physics.cpp:1364-1413); false = the backend reported an error (miBackendError)
bool physicsStepMI355X(game_scene& scene, memory_arena& arena, float& timer, const physics_settings& settings, float dt);
// the one-line hook of scene_entity::addComponent (src/scene/scene.h:35-112): the physics topology of this registry changed
void miOnPhysicsComponentChanged(entt::registry* registry);
// the library world that mirrors `scene` (synchronised with the ECS first); for diagnostics / tests
mi_world* miBackendWorld(game_scene& scene);
const char* miBackendError(game_scene& scene);
