// oracle_backend.cpp — TEST INFRASTRUCTURE.  The entry points of include/mi_physics.h that the backend stub (physics_mi355x.cpp) and its driver call,
// forwarded to the CPU oracle's ABI (oracle/_build/liboracle.so, canonical order = what the GPU runs, bit for bit): lets the CPU test suite run the
// compiled reference-side binding without a GPU (tests/test_reference_pin.py).  Never part of the product.
#include "mi_physics.h"
#include "mi_constraints.h"
extern "C" {
struct World;
int ora_world_create(int order_mode, World** out); void ora_world_destroy(World* w);
int ora_entity_create(World*, const mi_entity_desc*, uint32_t*); int ora_entity_set_force(World*, uint32_t, const float*);
int ora_collider_add(World*, uint32_t, const mi_collider_desc*, uint32_t*);
int ora_constraint_create(World*, uint32_t, uint32_t, uint32_t, const void*, uint32_t, uint32_t*); int ora_constraint_update(World*, uint32_t, uint32_t, const void*, uint32_t);
int ora_world_set_body_states(World*, uint32_t, const uint32_t*, const float*); int ora_entities_apply_forces(World*, uint32_t, const uint32_t*, const float*, const float*);
int ora_world_step(World*, const mi_step_settings*, float); int ora_world_num_entities(World*, uint32_t*);
int ora_world_get_transforms(World*, float*, float*, uint32_t); int ora_world_get_physics_transforms(World*, float*, float*, uint32_t); int ora_world_get_velocities(World*, float*, float*, uint32_t);
int ora_world_get_counts(World*, mi_step_counts*); int ora_debug_set_sweep_axis(World*, uint32_t); int ora_debug_set_solve_order(World*, const uint32_t*, uint32_t);

#define W(w) reinterpret_cast<World*>(w)
MI_API const char* mi_last_error(void) { return "(oracle backend: status code only)"; }
MI_API int mi_world_create(const mi_world_desc*, mi_world** out) { return ora_world_create(1 /* ORDER_CANONICAL */, reinterpret_cast<World**>(out)); }
MI_API void mi_world_destroy(mi_world* w) { ora_world_destroy(W(w)); }
MI_API int mi_entity_create(mi_world* w, const mi_entity_desc* d, uint32_t* out) { return ora_entity_create(W(w), d, out); }
MI_API int mi_entity_set_force(mi_world* w, uint32_t e, const float* f) { return ora_entity_set_force(W(w), e, f); }
MI_API int mi_collider_add(mi_world* w, uint32_t e, const mi_collider_desc* d, uint32_t* out) { return ora_collider_add(W(w), e, d, out); }
MI_API int mi_constraint_create(mi_world* w, uint32_t t, uint32_t a, uint32_t b, const void* pod, uint32_t bytes, uint32_t* out) { return ora_constraint_create(W(w), t, a, b, pod, bytes, out); }
MI_API int mi_constraint_update(mi_world* w, uint32_t t, uint32_t id, const void* pod, uint32_t bytes) { return ora_constraint_update(W(w), t, id, pod, bytes); }
MI_API int mi_world_set_body_states(mi_world* w, uint32_t n, const uint32_t* e, const float* s) { return ora_world_set_body_states(W(w), n, e, s); }
MI_API int mi_entities_apply_forces(mi_world* w, uint32_t n, const uint32_t* e, const float* f, const float* t) { return ora_entities_apply_forces(W(w), n, e, f, t); }
MI_API int mi_world_step(mi_world* w, const mi_step_settings* s, float dt) { return ora_world_step(W(w), s, dt); }
MI_API int mi_world_num_entities(mi_world* w, uint32_t* out) { return ora_world_num_entities(W(w), out); }
MI_API int mi_world_get_transforms(mi_world* w, float* p, float* r, uint32_t cap) { return ora_world_get_transforms(W(w), p, r, cap); }
MI_API int mi_world_get_physics_transforms(mi_world* w, float* p, float* r, uint32_t cap) { return ora_world_get_physics_transforms(W(w), p, r, cap); }
// (no device, no pinned rows to view: the stub falls back to the copying calls)
MI_API int mi_world_view_transforms(mi_world*, const float**, const float**, uint32_t*) { return MI_ERR_UNSUPPORTED; }
MI_API int mi_world_view_physics_transforms(mi_world*, const float**, const float**, uint32_t*) { return MI_ERR_UNSUPPORTED; }
MI_API int mi_world_view_velocities(mi_world*, const float**, const float**, uint32_t*) { return MI_ERR_UNSUPPORTED; }
MI_API int mi_world_get_velocities(mi_world* w, float* l, float* a, uint32_t cap) { return ora_world_get_velocities(W(w), l, a, cap); }
MI_API int mi_world_get_counts(mi_world* w, mi_step_counts* out) { return ora_world_get_counts(W(w), out); }
MI_API int mi_debug_set_sweep_axis(mi_world* w, uint32_t axis) { return ora_debug_set_sweep_axis(W(w), axis); }
MI_API int mi_debug_set_solve_order(mi_world* w, const uint32_t* pairs, uint32_t n) { return ora_debug_set_solve_order(W(w), pairs, n); }
}
