/*
 * mi_physics.h — C ABI of the MI355X-native rigid-body stepper.
 *
 * This is the drop-in boundary for the `src/physics` step path of pkurth/D3D12Renderer.
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repo root).  Plain pointers and sizes only; no C++ / torch types.  All functions
 * return MI_OK (0) or a negative mi_status; nothing throws across this boundary.
 *
 * Threading: one host thread per world.  All device work runs on a world-owned HIP stream;
 * every call returns after device completion.
 *
 * Index conventions (they matter for result parity):
 *   - rigid bodies are indexed in creation order (EnTT dense storage slot,
 *     src/physics/physics.cpp:654,1269-1273);
 *   - colliders are world-indexed in REVERSE creation order (EnTT iterates back to front,
 *     src/physics/physics.cpp:635-641);
 *   - a collider without a rigid body ("static collider") uses the dummy body index N_b
 *     (src/physics/physics.cpp:1214,1279).
 */
#ifndef MI_PHYSICS_H
#define MI_PHYSICS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_API __attribute__((visibility("default")))

typedef enum mi_status {
    MI_OK = 0,
    MI_ERR_INVALID_ARGUMENT = -1,
    MI_ERR_NO_DEVICE = -2,        /* HIP device / kernels unavailable: the product path never falls back to CPU */
    MI_ERR_OUT_OF_MEMORY = -3,
    MI_ERR_DEVICE = -4,           /* a HIP call failed; see mi_last_error() */
    MI_ERR_CAPACITY = -5,
    MI_ERR_UNSUPPORTED = -6
} mi_status;

/* collider_type — src/physics/physics.h:59-70 (order is load-bearing: narrow-phase buckets). */
typedef enum mi_collider_type {
    MI_COLLIDER_SPHERE = 0,
    MI_COLLIDER_CAPSULE = 1,
    MI_COLLIDER_CYLINDER = 2,
    MI_COLLIDER_AABB = 3,
    MI_COLLIDER_OBB = 4,
    MI_COLLIDER_HULL = 5,
    MI_COLLIDER_TYPE_COUNT = 6
} mi_collider_type;

/* physics_object_type — src/physics/physics.h:49-57. */
typedef enum mi_object_type {
    MI_OBJECT_RIGID_BODY = 0,
    MI_OBJECT_STATIC_COLLIDER = 1,
    MI_OBJECT_FORCE_FIELD = 2,
    MI_OBJECT_TRIGGER = 3
} mi_object_type;

/* constraint_type — src/physics/constraints.h:14-28. */
typedef enum mi_constraint_type {
    MI_CONSTRAINT_DISTANCE = 0,
    MI_CONSTRAINT_BALL = 1,
    MI_CONSTRAINT_FIXED = 2,
    MI_CONSTRAINT_HINGE = 3,
    MI_CONSTRAINT_CONE_TWIST = 4,
    MI_CONSTRAINT_SLIDER = 5,
    MI_CONSTRAINT_TYPE_COUNT = 6
} mi_constraint_type;

enum { MI_ENTITY_DESTROYED = 0xFFFFFFFFu };   /* kind of an entity slot after mi_entity_destroy (read-back only) */
enum { MI_ENTITY_DYNAMIC = 0, MI_ENTITY_KINEMATIC = 1, MI_ENTITY_STATIC = 2,
       MI_ENTITY_TRIGGER = 3,        /* transform + trigger_component (src/physics/physics.h:200-203): colliders report enter / leave */
       MI_ENTITY_FORCE_FIELD = 4 };  /* transform + force_field_component (src/physics/physics.h:182-185): see mi_entity_set_force */

/*
 * An entity = transform_component (+ optional rigid_body_component).
 * Replaces: scene.createEntity().addComponent<transform_component>(pos, rot)
 *           .addComponent<rigid_body_component>(kinematic, gravityFactor, linearDamping, angularDamping)
 * (src/scene/scene.h:35-112, src/physics/rigid_body.cpp:6-27).  kind == MI_ENTITY_STATIC creates
 * the transform only: its colliders become static colliders.
 */
typedef struct mi_entity_desc {
    float position[3];
    float rotation[4];          /* quaternion x,y,z,w (src/core/math.h:292-298) */
    float linear_velocity[3];
    float angular_velocity[3];
    float gravity_factor;       /* default 1 */
    float linear_damping;       /* default 0.4 (src/physics/rigid_body.h:21) */
    float angular_damping;      /* default 0.4 */
    uint32_t kind;              /* MI_ENTITY_* */
} mi_entity_desc;

/*
 * collider_component::asSphere/asCapsule/asCylinder/asAABB/asOBB/asHull + physics_material
 * (src/physics/physics.h:108-171, 40-47).  Shapes are in the entity's local space.
 *   sphere   : shape = {cx,cy,cz, r}
 *   capsule  : shape = {ax,ay,az, bx,by,bz, r}
 *   cylinder : shape = {ax,ay,az, bx,by,bz, r}
 *   aabb     : shape = {minx,miny,minz, maxx,maxy,maxz}
 *   obb      : shape = {qx,qy,qz,qw, cx,cy,cz, rx,ry,rz}
 *   hull     : shape = {qx,qy,qz,qw, px,py,pz}, hull_geometry = id from mi_hull_geometry_create
 */
typedef struct mi_collider_desc {
    uint32_t type;              /* mi_collider_type */
    uint32_t object_type;       /* MI_OBJECT_RIGID_BODY (resolved from the entity) — force_field / trigger reserved */
    float shape[12];
    uint32_t hull_geometry;
    float restitution;
    float friction;
    float density;
} mi_collider_desc;

/* physics_settings minus callbacks — src/physics/physics.h:382-400. */
typedef struct mi_step_settings {
    uint32_t fixed_frame_rate;              /* bool, default 1 */
    uint32_t frame_rate;                    /* default 120 */
    uint32_t max_physics_iterations_per_frame; /* default 4 */
    uint32_t num_rigid_solver_iterations;   /* default 30 */
} mi_step_settings;

/* The bit-exact integers of one internal step (CPU_PROFILE_STAT names, src/physics/physics.cpp:1258-1262). */
typedef struct mi_step_counts {
    uint32_t num_rigid_bodies;
    uint32_t num_colliders;
    uint32_t num_broadphase_overlaps;
    uint32_t num_collisions;    /* manifolds */
    uint32_t num_contacts;
    uint32_t num_colors;        /* contact colours used by the solver schedule */
    uint32_t sorting_axis;      /* SAP axis used this step (src/physics/collision_broad.cpp:345) */
    uint32_t reserved;          /* product library: contact-solve kernel launches of the step */
} mi_step_counts;

/* collision_contact + constraint_body_pair + collider_pair — src/physics/physics.h:347-354. */
typedef struct mi_contact {
    float point[3];
    float penetration_depth;
    float normal[3];
    uint32_t friction_restitution;  /* 16:16 fixed point (src/physics/collision_narrow.cpp:2235-2238) */
    uint32_t collider_a, collider_b;
    uint32_t body_a, body_b;
} mi_contact;

/* Per-stage device time of the last internal step, milliseconds (profiler block names of SURVEY §5). */
typedef struct mi_stage_times {
    float world_colliders;
    float broadphase;
    float narrowphase;
    float integrate_forces;
    float schedule;
    float init_constraints;
    float solve;
    float integrate_velocities;
    float total;
} mi_stage_times;

typedef struct mi_world_desc {
    int32_t device;             /* HIP device ordinal */
    uint32_t flags;             /* reserved, 0 */
} mi_world_desc;

typedef struct mi_world mi_world;

/* Library */
MI_API const char* mi_last_error(void);
MI_API int mi_version(void);

/* physics_world ctor/dtor  <->  game_scene + memory_arena::initialize (src/physics/physics.cpp:1205). */
MI_API int mi_world_create(const mi_world_desc* desc, mi_world** out_world);
MI_API void mi_world_destroy(mi_world* world);

/* addRigidBody  <->  createEntity().addComponent<transform_component>()...addComponent<rigid_body_component>() */
MI_API int mi_entity_create(mi_world* world, const mi_entity_desc* desc, uint32_t* out_entity);
MI_API int mi_entities_create(mi_world* world, uint32_t count, const mi_entity_desc* descs, uint32_t* out_first_entity);
/*
 * game_scene::deleteEntity (src/scene/scene.cpp:124-150): the entity's colliders (removeColliderFromBroadphase,
 * src/physics/collision_broad.cpp:62-75), its constraints (deleteAllConstraintsFromEntity) and its rigid body / trigger / force field
 * go.  EnTT pool semantics decide the order everything else is processed in afterwards: in every component pool the LAST element
 * moves into the freed slot (rigid bodies, colliders — hence collider ids reported by events are pool positions and change for the
 * moved collider —, triggers, force fields).  Entity ids stay valid; the destroyed id is never reused.  The colour history and the
 * previous-step collision / trigger lists (which are keyed by pool positions) restart, so the next step reports every live
 * collision as begun again.
 */
MI_API int mi_entity_destroy(mi_world* world, uint32_t entity);
/* addComponent<collider_component>(collider_component::asXxx(shape, material)) (src/scene/scene.h:38-65). */
MI_API int mi_collider_add(mi_world* world, uint32_t entity, const mi_collider_desc* desc, uint32_t* out_collider);
MI_API int mi_colliders_add(mi_world* world, uint32_t count, const uint32_t* entities, const mi_collider_desc* descs);
/* allocateBoundingHullGeometry (src/physics/physics.cpp:58-84) / bounding_hull_geometry::fromMesh. */
MI_API int mi_hull_geometry_create(mi_world* world, const float* vertices_xyz, uint32_t num_vertices,
                                   const uint32_t* triangles_abc, uint32_t num_triangles, uint32_t* out_geometry);

/*
 * addConstraint(a, b, const xxx_constraint&) (src/physics/physics.h:235-240).  `pod` is laid out like the
 * reference structs (src/physics/constraints.h:73-80,129-135,175-183,229-257,346-380,497-520) with
 * 4-byte packing; see mi_*_constraint in mi_constraints.h.
 */
MI_API int mi_constraint_create(mi_world* world, uint32_t type, uint32_t entity_a, uint32_t entity_b,
                                const void* pod, uint32_t pod_bytes, uint32_t* out_constraint);
/* getConstraint(scene, handle) = mutable access (motors/limits) (src/physics/physics.h:244-249). */
MI_API int mi_constraint_update(mi_world* world, uint32_t type, uint32_t constraint, const void* pod, uint32_t pod_bytes);
MI_API int mi_constraint_get(mi_world* world, uint32_t type, uint32_t constraint, void* pod, uint32_t pod_bytes);
/* mi_constraint_update for `count` constraints of one type (pods = count consecutive PODs of pod_bytes each): a policy writing the
 * motor targets of thousands of ragdolls per step.  Updates only re-send the POD arrays; the step stays on its fast path. */
MI_API int mi_constraints_update(mi_world* world, uint32_t type, uint32_t count, const uint32_t* constraints, const void* pods, uint32_t pod_bytes);
/* deleteConstraint(scene, handle), deleteAllConstraints(scene), deleteAllConstraintsFromEntity(entity)
 * (src/physics/physics.h:251-260, physics.cpp:443-539).  Constraint ids stay valid across deletions of other constraints; the
 * solver's pool order follows EnTT's swap-and-pop (the last constraint of the type moves into the freed slot). */
MI_API int mi_constraint_destroy(mi_world* world, uint32_t type, uint32_t constraint);
MI_API int mi_constraints_destroy_all(mi_world* world);
MI_API int mi_entity_destroy_constraints(mi_world* world, uint32_t entity);
/* add{Distance,Ball,Fixed,Hinge,ConeTwist,Slider}ConstraintFromGlobalPoints (src/physics/physics.cpp:128-333). */
MI_API int mi_constraint_create_from_global(mi_world* world, uint32_t type, uint32_t entity_a, uint32_t entity_b,
                                            const float* global_anchor, const float* global_axis,
                                            float limit_min_or_swing, float limit_max_or_twist, uint32_t* out_constraint);

/* force_field_component::force (src/physics/physics.h:182-185) of a MI_ENTITY_FORCE_FIELD entity, in the entity's local frame
 * (getForceFieldStates rotates it by the transform, src/physics/physics.cpp:759-787).  With colliders the field is localized:
 * every overlap of one of its colliders with a rigid body's collider adds the force to that body for the step
 * (handleNonCollisionInteractions, physics.cpp:952-970); without colliders it is global and acts on every rigid body. */
MI_API int mi_entity_set_force(mi_world* world, uint32_t entity, const float* force3);

/*
 * Heightmap terrain — heightmap_collider_component (src/terrain/heightmap_collider.h:126-151), collided by heightmapCollision
 * (src/physics/heightmap_collision.cpp:509-618) right after the narrow phase, into the same contact arrays.  One per world.
 *   mi_heightmap_create            = heightmap_collider_component(chunksPerDim, chunkSize, material)
 *   mi_heightmap_set_chunk_heights = collider(x, z).setHeights(heights): 129 x 129 uint16, row-major [z][x]
 *                                    (TERRAIN_LOD_0_VERTICES_PER_DIMENSION); chunks without heights collide with nothing
 *   mi_heightmap_update            = update(minCorner, amplitudeScale): height = minCorner.y + h / 65535 * amplitudeScale
 *   mi_heightmap_get_height        = getHeightAt(coord): bilinear height, -FLT_MAX outside
 * Sphere, capsule, AABB and OBB colliders of rigid bodies collide with the terrain triangles (one contact per touched
 * triangle plus the lowest-point contact, at most 255 per collider); cylinders and hulls do not (the reference reads an
 * uninitialised point for them).  mi_contact::collider_b of a terrain contact is 0xFFFFFFFF and its body_b is the static dummy.
 */
MI_API int mi_heightmap_create(mi_world* world, uint32_t chunks_per_dim, float chunk_size, float restitution, float friction);
MI_API int mi_heightmap_set_chunk_heights(mi_world* world, uint32_t chunk_x, uint32_t chunk_z, const uint16_t* heights129x129);
MI_API int mi_heightmap_update(mi_world* world, const float* min_corner3, float amplitude_scale);
MI_API int mi_heightmap_get_height(mi_world* world, float x, float z, float* out_height);

/* rb.forceAccumulator += f; rb.torqueAccumulator += tau (src/physics/physics.cpp:623-627). */
MI_API int mi_entity_apply_force(mi_world* world, uint32_t entity, const float* force3, const float* torque3);
/* The same for many rigid bodies at once, in order (either array may be NULL); added on the device when no topology edit is pending. */
MI_API int mi_entities_apply_forces(mi_world* world, uint32_t count, const uint32_t* entities, const float* forces3, const float* torques3);
/* testPhysicsInteraction(scene, ray, strength) (src/physics/physics.h:404, physics.cpp:555-629) for `count` rays, applied in
 * order: the closest rigid-body collider along ray i (ray::intersectSphere/Capsule/Cylinder/AABB/OBB/Hull in the entity's
 * physics_transform1 frame) receives force = direction * strength at the hit point.  strengths NULL = 1000 (the reference's
 * default); entity_ranges2 NULL = the whole scene, otherwise ray i only sees colliders of entities [lo_i, hi_i) — batched
 * environments in one world. */
MI_API int mi_world_test_interactions(mi_world* world, uint32_t count, const float* origins3, const float* directions3,
                                      const float* strengths, const uint32_t* entity_ranges2);

/* physicsStep(scene, arena, timer, settings, dt) (src/physics/physics.cpp:1364-1413). */
MI_API int mi_world_step(mi_world* world, const mi_step_settings* settings, float dt);
/* n × physicsStepInternal(scene, arena, settings, dt) (src/physics/physics.cpp:1180-1362); no interpolation. */
MI_API int mi_world_step_fixed(mi_world* world, const mi_step_settings* settings, float dt, uint32_t num_steps);

/* One internal step with a HIP event pair around every contact-solver launch (the dominant kernel: k_contact_solve_flow,
 * or k_contact_solve per colour with MI_SOLVER=launch): returns the number of launches, their summed device time and the
 * contact updates (contacts x iterations) they performed. */
MI_API int mi_world_step_profiled(mi_world* world, const mi_step_settings* settings, float dt, uint32_t* out_launches,
                                  float* out_kernel_ms, uint64_t* out_contact_updates);

/* Stepping statistics since world creation: internal steps taken, steps run speculatively (one host read-back at the end,
 * sizes bounded from the previous step) and how many of those had to be re-run synchronously because a bound was exceeded. */
MI_API int mi_world_get_step_mode_stats(mi_world* world, uint32_t* out_steps, uint32_t* out_speculative, uint32_t* out_retries);

/* Read-back (transform_component / rigid_body_component fields), entity order. */
MI_API int mi_world_num_entities(mi_world* world, uint32_t* out);
MI_API int mi_world_get_transforms(mi_world* world, float* positions_xyz, float* rotations_xyzw, uint32_t capacity);
MI_API int mi_world_get_physics_transforms(mi_world* world, float* positions_xyz, float* rotations_xyzw, uint32_t capacity);
/* The same values without the last host copy, for a caller that reads every entity's transform after every step (the renderer
 * reading transform_component, src/physics/physics.cpp:1392-1411): pointers to the library's pinned host rows, [count][3] positions
 * and [count][4] rotations.  The rows are produced on the device in this layout and cross the bus as ONE copy which, once a caller
 * has asked after a step, every later step enqueues itself before it returns.  Valid until the SECOND next stepping call on this
 * world (two sets alternate) or until entities are added; read-only.  MI_ERR_UNSUPPORTED when the poses are not coming from the device right now (nothing
 * stepped since the last full download, a topology change is pending, a sharded world): mi_world_get_transforms covers every case. */
MI_API int mi_world_view_transforms(mi_world* world, const float** positions_xyz, const float** rotations_xyzw, uint32_t* out_count);
MI_API int mi_world_view_physics_transforms(mi_world* world, const float** positions_xyz, const float** rotations_xyzw, uint32_t* out_count);
/* The same rows WITHOUT waiting for a copy that is still on the bus: the NEWEST set that is complete in host memory — the last step's if its copy has landed, else the step's before
 * (*out_of_step = the internal step whose state they are; mi_world_get_step_mode_stats counts the same steps).  For a renderer that accepts one frame of latency — the
 * reference's own loop interpolates between the two last poses anyway (src/physics/physics.cpp:1395-1412): its next physicsStep then runs on the device while the rows of the
 * step it has not seen yet cross the bus, and a frame costs max(step, copy) instead of their sum.  The call never blocks except for the very first rows of a world.  What it
 * hands out stays intact until a stepping call produces into the same set: the rows of the step before the last are overwritten by the NEXT stepping call.  Read-only. */
MI_API int mi_world_view_transforms_landed(mi_world* world, uint32_t physics_transforms, const float** positions_xyz, const float** rotations_xyzw, uint32_t* out_count, uint64_t* out_of_step);
/* ... and the linear / angular velocities of every entity's rigid body (rigid_body_component::linearVelocity / angularVelocity; zero for an entity without one),
 * [count][3] each, out of the same rows: they ride along from the frame after the first request for velocities on (mi_world_get_velocities takes them from there too). */
MI_API int mi_world_view_velocities(mi_world* world, const float** linear_xyz, const float** angular_xyz, uint32_t* out_count);
MI_API int mi_world_get_velocities(mi_world* world, float* linear_xyz, float* angular_xyz, uint32_t capacity);
MI_API int mi_world_get_mass_properties(mi_world* world, float* inv_mass, float* inv_inertia_9, float* local_cog_xyz, uint32_t capacity);
MI_API int mi_world_get_counts(mi_world* world, mi_step_counts* out);
/* Contacts of the last internal step in solver (canonical) order; returns count in *out_count. */
MI_API int mi_world_get_contacts(mi_world* world, mi_contact* out, uint32_t capacity, uint32_t* out_count);
/* Diagnostics (no device needed): the tile -> XCD assignment of the XCD-partitioned contact solver for tile `tile_in_bin` of a
 * schedule bin with `tiles_in_bin` tiles: out[0] = owning XCD, out[1] = its rank inside that XCD's share of the bin,
 * out[2] = size of `query_xcd`'s share of the bin. */
MI_API int mi_debug_tile_owner(uint32_t tile_in_bin, uint32_t tiles_in_bin, uint32_t bin, uint32_t query_xcd, uint32_t* out);
/* Per-stage device times of the last internal step (HIP events on the world's stream).  Nothing is timed by default — even events attached to a
 * kernel's dispatch leave the device idle for a few microseconds (the solver's start / stop pair: ~11 us of a 1 ms step).  level 2: the whole step
 * (`total`) and the solve stage; level 3: the solve stage alone (`total` stays 0: the step's own start / stop events are two more such gaps — what bench.py runs its timed region
 * with, its roofline needs the solver launch's duration only); level 1: every stage (an event pair each); level 0: off. */
MI_API int mi_world_set_stage_timing(mi_world* world, uint32_t level);
MI_API int mi_world_get_stage_times(mi_world* world, mi_stage_times* out);
/* Contact-solver kernel of the last internal step: 0 k_contact_solve (a launch per colour per sweep), 1 k_contact_solve_flow,
 * 2 k_contact_solve_persist (default without joints), 3 k_solve_flow_islands (contacts + joint islands in one launch),
 * 4 k_contact_solve_persist with the tiles partitioned over the XCDs (default from 16384 manifolds up), 5 the same kernel with all
 * tiles on ONE XCD (default below that: every body hand-over through one L2). */
MI_API int mi_world_get_solver_kind(mi_world* world, uint32_t* out_kind);
/* Diagnostics: steps of a small scene whose enqueued work is bit-identical to that of a captured step are replayed as ONE HIP graph
 * (HIP runtime >= 7.2; MI_GRAPH=0 / force / all).  out[0] = enabled, out[1] = steps replayed, out[2] = graphs captured,
 * out[3] = speculative steps launched plainly. */
MI_API int mi_debug_step_graph_stats(mi_world* world, uint32_t* out4);
/* Tests: in how many valid steps the colouring was finished by the tail inside k_bin_hist (the colouring rounds the host enqueued — what the previous step needed + 1 — left
 * manifolds uncoloured), and how many rounds ran there in total.  No reference counterpart (scheduleConstraintsSIMD is one serial pass, src/physics/constraints.cpp:51-184). */
MI_API int mi_debug_color_tail_stats(mi_world* world, uint64_t* out_steps, uint64_t* out_rounds);
/* Tests: how many times the pose rows (mi_world_view_transforms) were enqueued by a step itself, and how many times only when asked. */
MI_API int mi_debug_pose_stream_stats(mi_world* world, uint32_t* out_ahead, uint32_t* out_on_demand);
/* Step-ahead (no reference counterpart; what it takes off the path is the host's time between two physicsStep calls, src/physics/physics.cpp:1364-1413): a speculative step
 * enqueues the NEXT step's first kernel (world colliders + broad-phase classification) behind its own end-of-step record, into a second set of world-shape / AABB rows; the next step
 * adopts the result unless something its inputs depend on changed in between (a state write, an upload, another step mode, a void step).  How often it was enqueued / adopted. */
MI_API int mi_debug_step_ahead_stats(mi_world* world, uint64_t* out_enqueued, uint64_t* out_adopted);
/* Sum of the per-stage device times and of the contact updates (contacts x solver iterations) over the internal steps since
 * the last reset (so a benchmark loop does not have to call back into the library after every step). */
MI_API int mi_world_get_accumulated_stage_times(mi_world* world, mi_stage_times* out_sum, uint32_t* out_steps,
                                                uint64_t* out_contact_updates, uint32_t reset);
/* Stage dumps for parity bisecting: world AABBs (6 floats per collider, world index order) and the
 * solver colour of every manifold of the last step. */
MI_API int mi_world_get_aabbs(mi_world* world, float* out_min_max6, uint32_t capacity);
MI_API int mi_world_get_manifold_colors(mi_world* world, uint32_t* out_colors, uint32_t capacity);
/* Parity against the reference's own constraint order (debug speed: everything below runs sequentially).
 * PGS is order dependent; the library's canonical order is colour-major, the reference solves, per iteration, the joints of
 * every type in pool order and then the contacts in emission order (src/physics/constraints.cpp:3748-3770, 3381-3449).  These two calls make
 * the NEXT internal step follow what the caller says the reference did, so that a test can hold the library against the reference itself,
 * bit for bit, step after step:
 *   mi_debug_set_sweep_axis    the axis that step sweeps along (the reference picks it from float sums in pool order,
 *                              collision_broad.cpp:376-384, 443-444 — the library from an order-free integer statistic; they agree except
 *                              in near ties); later steps choose their own again
 *   mi_debug_set_solve_order   `pairs` = the oriented collider pairs (world indices A, B: 2 * count values) of the step's contact manifolds
 *                              in the order the reference emitted them.  The step then solves exactly these manifolds one after the other
 *                              in that order (one lane), the joints of each type one after the other in pool order, instead of the coloured
 *                              schedule; a manifold the step finds that is not in the list (or a listed one it does not find) makes the
 *                              step fail with MI_ERR_INVALID_ARGUMENT.  The list also ORIENTS pairs of equal shape type: where the AABB starts of
 *                              such a pair tie exactly on the sweep axis, the reference's (A, B) follows the history of its persistent endpoint
 *                              array (stable insertion sort, collision_broad.cpp:386-398), which no rule of the current state reproduces; a pair
 *                              listed the other way round is turned before the narrow phase.  Not with heightmap terrain or sharding
 *                              (MI_ERR_UNSUPPORTED). */
MI_API int mi_debug_set_sweep_axis(mi_world* world, uint32_t axis);
MI_API int mi_debug_set_solve_order(mi_world* world, const uint32_t* pairs, uint32_t count);
/*   mi_debug_set_solve_dataflow (sticky switch)  a step that follows a caller's order runs it through the PRODUCTION contact solver instead of the one-lane kernel: every
 *                              manifold is coloured with its level in the order's dependency graph (1 + the highest level among the earlier manifolds of its bodies),
 *                              so the colour-major dataflow schedule — bins, tiles, version bookkeeping, k_contact_solve_persist's tile code — performs the caller's
 *                              sequence of updates on every body: the caller's sequential result, bit for bit.  Needs an order at most 64 levels deep and no joints;
 *                              otherwise that step uses the one-lane kernel as before.  mi_debug_solve_order_depth: the depth the last such step ran with (0: it did not).
 *                              The reference's per-contact update it must reproduce: src/physics/constraints.cpp:3381-3449, its order 3748-3770. */
MI_API int mi_debug_set_solve_dataflow(mi_world* world, uint32_t enable);
MI_API int mi_debug_solve_order_depth(mi_world* world, uint32_t* out_depth);

/*
 * Cloth — cloth_component (src/physics/cloth.h:5-60, cloth.cpp): a gridSizeX x gridSizeY particle grid (upper row fixed) with
 * stretch / shear / bend distance constraints, stepped after the rigid bodies of every internal step
 * (physics.cpp:1352-1358): wind from the global force field, gravity, then numClothVelocityIterations /
 * numClothPositionIterations / numClothDriftIterations Gauss-Seidel passes (physics_settings defaults 0 / 1 / 0,
 * src/physics/physics.h:390-392; mi_world_set_cloth_iterations changes them).  Cloth does not interact with rigid bodies.
 *   mi_cloth_create             = cloth_component(width, height, gridSizeX, gridSizeY, totalMass, stiffness, damping, gravityFactor)
 *   mi_cloth_set_fixed_vertices = setWorldPositionOfFixedVertices(transform, moveRigid)
 *   mi_cloth_set_properties     = writing totalMass / stiffness / damping / gravityFactor (recalculateProperties on the next step)
 */
typedef struct mi_cloth_desc {
    float width, height;
    uint32_t grid_size_x, grid_size_y;
    float total_mass;
    float stiffness;        /* default 0.5 */
    float damping;          /* default 0.3 */
    float gravity_factor;   /* default 1 */
} mi_cloth_desc;
MI_API int mi_cloth_create(mi_world* world, const mi_cloth_desc* desc, uint32_t* out_cloth);
MI_API int mi_cloth_set_fixed_vertices(mi_world* world, uint32_t cloth, const float* position3, const float* rotation4, uint32_t move_rigid);
MI_API int mi_cloth_set_properties(mi_world* world, uint32_t cloth, float total_mass, float stiffness, float damping, float gravity_factor);
MI_API int mi_cloth_get_state(mi_world* world, uint32_t cloth, float* out_positions3, float* out_velocities3, uint32_t capacity_particles);
MI_API int mi_world_set_cloth_iterations(mi_world* world, uint32_t velocity_iterations, uint32_t position_iterations, uint32_t drift_iterations);

/*
 * Checkpoint / resume of the solver-relevant state: body states and accumulators, physics_transform0 and the
 * interpolated transforms, the step accumulator, the SAP axis of the next step, the contact-colour history (= the previous
 * step's collision list), the previous trigger overlaps, the constraint PODs.  A world built from the same scene description
 * that loads the blob continues bit-identically to the world that saved it.  save: out == NULL only reports the size.
 */
MI_API int mi_world_save_checkpoint(mi_world* world, void* out, uint64_t capacity, uint64_t* out_size);
MI_API int mi_world_load_checkpoint(mi_world* world, const void* data, uint64_t size);

/*
 * Collision events — collisionBeginCallback / collisionEndCallback of physics_settings (src/physics/physics.h:398-399),
 * fired by handleCollisionCallbacks (src/physics/physics.cpp:1041-1178).  The reference calls std::function callbacks
 * synchronously inside the step; across a C ABI they are polled: events of the internal steps since the last poll, per step
 * in ascending (collider_a, collider_b) order — the order of the reference's sorted merge of the previous and the current
 * frame's collision lists.  Like there, a pair is identified by its ORIENTED collider pair (a re-oriented pair ends and begins).
 * Disabled by default (the reference only diffs the lists when a callback is set); costs nothing when disabled.
 */
typedef enum mi_event_type {
    MI_EVENT_COLLISION_BEGIN = 0, MI_EVENT_COLLISION_END = 1,
    /* trigger_event_enter / trigger_event_leave (src/physics/physics.h:187-198): entity_a = the trigger entity, entity_b = the rigid
     * body's entity; de-duplicated per entity pair (several colliders may overlap), fired in ascending (trigger, body) order before
     * the collision events of the step, like handleNonCollisionInteractions (src/physics/physics.cpp:952-1039); colliders = ~0. */
    MI_EVENT_TRIGGER_ENTER = 2, MI_EVENT_TRIGGER_LEAVE = 3
} mi_event_type;
typedef struct mi_event {
    uint32_t type;                  /* mi_event_type */
    uint32_t entity_a, entity_b;    /* the colliders' parent entities (collision_begin_event::entityA/B) */
    uint32_t collider_a, collider_b;/* collider ids as returned by mi_collider_add (creation order) */
    float point[3];                 /* begin only: mean contact point, mean normal, velocity of B relative to A at the point */
    float normal[3];
    float relative_velocity[3];
} mi_event;
MI_API int mi_world_enable_events(mi_world* world, uint32_t enable);
MI_API int mi_world_poll_events(mi_world* world, mi_event* out, uint32_t capacity, uint32_t* out_count);

/*
 * Ghost-region exchange support (multi-GPU spatial sharding).  A body state is 13 floats:
 * position[3], rotation[4] (x,y,z,w), linear_velocity[3], angular_velocity[3] — i.e. physics_transform1 +
 * rigid_body_component velocities.  The host variants take entity ids and host buffers; the *_device variants
 * take rigid-body indices and state buffers that already live in this world's device memory (e.g. the
 * data_ptr() of torch CUDA tensors used with RCCL), so the exchange never round-trips through the host.
 */
#define MI_BODY_STATE_FLOATS 13
MI_API int mi_world_get_body_states(mi_world* world, uint32_t count, const uint32_t* entities, float* out_states13);
MI_API int mi_world_set_body_states(mi_world* world, uint32_t count, const uint32_t* entities, const float* states13);
MI_API int mi_world_entities_to_bodies(mi_world* world, uint32_t count, const uint32_t* entities, uint32_t* out_body_indices);
MI_API int mi_world_get_body_states_device(mi_world* world, uint32_t count, const uint32_t* body_indices_dev, float* out_states13_dev);
MI_API int mi_world_set_body_states_device(mi_world* world, uint32_t count, const uint32_t* body_indices_dev, const float* states13_dev);
/* The *_device calls without the host synchronisation: the copy kernels are only enqueued on the world's HIP stream
 * (mi_world_get_stream returns the hipStream_t).  Running the collective on that stream (torch.cuda.ExternalStream + RCCL) orders
 * gather -> exchange -> scatter -> next step on the device, with no host round trip in between. */
MI_API int mi_world_get_body_states_device_async(mi_world* world, uint32_t count, const uint32_t* body_indices_dev, float* out_states13_dev);
MI_API int mi_world_set_body_states_device_async(mi_world* world, uint32_t count, const uint32_t* body_indices_dev, const float* states13_dev);
MI_API int mi_world_get_stream(mi_world* world, void** out_hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_PHYSICS_H */
