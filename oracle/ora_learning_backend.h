// ora_learning_backend.h — TEST INFRASTRUCTURE.  The backend header the parity tests hand to d3d12renderer_amd/csrc/learning.cpp
// (-DMI_LEARNING_BACKEND_HEADER): the product's environment code compiled over the CPU oracle's C ABI (oracle/ora_world.h), so tests/test_learning.py
// runs identical environment code over both physics backends.  Never part of the product build.
#pragma once
#include <cstdlib>
#define PHYS(name) ora_##name
namespace ora { struct World; }
typedef ora::World phys_world;
extern "C" {
int ora_world_create(int order_mode, phys_world** out);
void ora_world_destroy(phys_world*);
int ora_entities_create(phys_world*, uint32_t, const mi_entity_desc*, uint32_t*);
int ora_colliders_add(phys_world*, uint32_t, const uint32_t*, const mi_collider_desc*);
int ora_constraint_create_from_global(phys_world*, uint32_t, uint32_t, uint32_t, const float*, const float*, float, float, uint32_t*);
int ora_constraint_get(phys_world*, uint32_t, uint32_t, void*, uint32_t);
int ora_constraints_update(phys_world*, uint32_t, uint32_t, const uint32_t*, const void*, uint32_t);
int ora_world_step(phys_world*, const mi_step_settings*, float);
int ora_world_get_transforms(phys_world*, float*, float*, uint32_t);
int ora_world_get_velocities(phys_world*, float*, float*, uint32_t);
int ora_world_get_mass_properties(phys_world*, float*, float*, float*, uint32_t);
int ora_world_set_body_states(phys_world*, uint32_t, const uint32_t*, const float*);
int ora_world_test_interactions(phys_world*, uint32_t, const float*, const float*, const float*, const uint32_t*);
}
// canonical order = the schedule the device runs; MI_LEARNING_ORACLE_ORDER=0 (tests): the reference's own order, to compare with oracle/_ref
static inline int physCreateWorld(int /*device*/, phys_world** out) { const char* om = std::getenv("MI_LEARNING_ORACLE_ORDER"); return ora_world_create(om ? std::atoi(om) : 1, out); }
static inline const char* physLastError() { return nullptr; }
