// ref_shim.cpp — TEST INFRASTRUCTURE.  extern "C" entry points over the REFERENCE's own physics code
// (compiled from /root/reference by oracle/refbuild/build_ref.py into oracle/_ref/libref.so).
//
// The exports mirror include/mi_physics.h with a `ref_` prefix (the subset a scene needs), so the ctypes class that drives
// the product and the oracle (d3d12renderer_amd/capi.py) drives the reference too and tests can step the same scene through
// all three.  Every function below only forwards to the reference's API:
//   scene.createEntity / addComponent<transform_component | rigid_body_component | collider_component | ...>   (src/scene/scene.h)
//   addConstraint / add*ConstraintFromGlobalPoints / getConstraint / deleteConstraint                            (src/physics/physics.h:215-260)
//   physicsStep                                                                                                  (src/physics/physics.cpp:1364)
// The per-step dumps (AABBs, broad-phase pairs, contacts) come from two instrumentation calls that build_ref.py inserts into
// the temporary copy of physicsStepInternal (ref_tap_broadphase / ref_tap_step): they read, never write.
#define private public      // cloth_component keeps its particles private; the shim only reads them (ref_cloth_get_state)
#include "physics/physics.h"
#undef private
#include "physics/collision_broad.h"
#include "scene/scene.h"
#include "terrain/heightmap_collider.h"
#include "mi_physics.h"
#include "mi_constraints.h"

// implemented in the patched copies (see build_ref.py)
uint32 ref_allocate_hull_geometry(vec3* vertices, uint32 numVertices, indexed_triangle16* triangles, uint32 numTriangles);
uint32 ref_sap_sorting_axis(game_scene& scene);

namespace {

struct ref_world
{
	game_scene scene;
	memory_arena arena;
	float timer = 0.f;
	physics_settings settings;

	std::vector<scene_entity> entities;      // by creation index (the ABI's entity ids)
	std::vector<uint32> kinds;
	std::vector<entity_handle> colliderEntities; // by collider creation index
	std::vector<entity_handle> constraintEntities[MI_CONSTRAINT_TYPE_COUNT];   // by constraint id (per type), null once deleted
	std::unordered_map<uint32, uint32> entityIdOfHandle;
	std::unordered_map<uint32, uint32> colliderIdOfHandle;
	scene_entity heightmapEntity;
	std::vector<scene_entity> cloths;
	std::vector<uint32> hullIds;              // world-local hull geometry id -> index into the reference's process-wide boundingHullGeometries
	bool eventsEnabled = false;
	std::vector<mi_event> events;

	// last-step dumps
	mi_step_counts counts{};
	std::vector<bounding_box> aabbs;
	std::vector<collider_pair> bpPairs;
	std::vector<mi_contact> contacts;
};

thread_local ref_world* g_stepping = nullptr;

vec3 v3(const float* f) { return vec3(f[0], f[1], f[2]); }
quat q4(const float* f) { return quat(f[0], f[1], f[2], f[3]); }
void put3(float* o, vec3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
void put4(float* o, quat q) { o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; }

collider_component colliderFromDesc(const ref_world* w, const mi_collider_desc& d)
{
	physics_material mat = { physics_material_type_none, d.restitution, d.friction, d.density };
	const float* f = d.shape;
	switch (d.type)
	{
		case MI_COLLIDER_SPHERE: return collider_component::asSphere(bounding_sphere{ v3(f), f[3] }, mat);
		case MI_COLLIDER_CAPSULE: return collider_component::asCapsule(bounding_capsule{ v3(f), v3(f + 3), f[6] }, mat);
		case MI_COLLIDER_CYLINDER: return collider_component::asCylinder(bounding_cylinder{ v3(f), v3(f + 3), f[6] }, mat);
		case MI_COLLIDER_AABB: return collider_component::asAABB(bounding_box{ v3(f), v3(f + 3) }, mat);
		case MI_COLLIDER_OBB: { bounding_oriented_box b; b.rotation = q4(f); b.center = v3(f + 4); b.radius = v3(f + 7); return collider_component::asOBB(b, mat); }
		default: { bounding_hull h; h.rotation = q4(f); h.position = v3(f + 4); h.geometryIndex = w->hullIds[d.hull_geometry]; return collider_component::asHull(h, mat); }
	}
}

// --- constraint PODs: member-by-member (the reference's structs carry 16-byte alignment padding, the ABI's do not)
void toRef(const mi_distance_constraint& s, distance_constraint& d) { d.localAnchorA = v3(s.local_anchor_a); d.localAnchorB = v3(s.local_anchor_b); d.globalLength = s.global_length; }
void toRef(const mi_ball_constraint& s, ball_constraint& d) { d.localAnchorA = v3(s.local_anchor_a); d.localAnchorB = v3(s.local_anchor_b); }
void toRef(const mi_fixed_constraint& s, fixed_constraint& d) { d.initialInvRotationDifference = q4(s.initial_inv_rotation_difference); d.localAnchorA = v3(s.local_anchor_a); d.localAnchorB = v3(s.local_anchor_b); }
void toRef(const mi_hinge_constraint& s, hinge_constraint& d)
{
	d.localAnchorA = v3(s.local_anchor_a); d.localAnchorB = v3(s.local_anchor_b);
	d.localHingeAxisA = v3(s.local_hinge_axis_a); d.localHingeAxisB = v3(s.local_hinge_axis_b);
	d.minRotationLimit = s.min_rotation_limit; d.maxRotationLimit = s.max_rotation_limit; d.maxMotorTorque = s.max_motor_torque;
	d.motorType = (constraint_motor_type)s.motor_type; d.motorVelocity = s.motor_velocity_or_target_angle;
	d.localHingeTangentA = v3(s.local_hinge_tangent_a); d.localHingeBitangentA = v3(s.local_hinge_bitangent_a); d.localHingeTangentB = v3(s.local_hinge_tangent_b);
}
void toRef(const mi_cone_twist_constraint& s, cone_twist_constraint& d)
{
	d.localAnchorA = v3(s.local_anchor_a); d.localAnchorB = v3(s.local_anchor_b);
	d.localLimitAxisA = v3(s.local_limit_axis_a); d.localLimitAxisB = v3(s.local_limit_axis_b);
	d.localLimitTangentA = v3(s.local_limit_tangent_a); d.localLimitBitangentA = v3(s.local_limit_bitangent_a); d.localLimitTangentB = v3(s.local_limit_tangent_b);
	d.swingLimit = s.swing_limit; d.twistLimit = s.twist_limit;
	d.swingMotorType = (constraint_motor_type)s.swing_motor_type; d.swingMotorVelocity = s.swing_motor_velocity_or_target_angle;
	d.maxSwingMotorTorque = s.max_swing_motor_torque; d.swingMotorAxis = s.swing_motor_axis;
	d.twistMotorType = (constraint_motor_type)s.twist_motor_type; d.twistMotorVelocity = s.twist_motor_velocity_or_target_angle;
	d.maxTwistMotorTorque = s.max_twist_motor_torque;
}
void toRef(const mi_slider_constraint& s, slider_constraint& d)
{
	d.initialInvRotationDifference = q4(s.initial_inv_rotation_difference); d.localAnchorA = v3(s.local_anchor_a); d.localAnchorB = v3(s.local_anchor_b);
	d.localAxisA = v3(s.local_axis_a); d.negDistanceLimit = s.neg_distance_limit; d.posDistanceLimit = s.pos_distance_limit;
	d.maxMotorForce = s.max_motor_force; d.motorType = (constraint_motor_type)s.motor_type; d.motorVelocity = s.motor_velocity_or_target_distance;
}
void fromRef(const distance_constraint& s, mi_distance_constraint& d) { put3(d.local_anchor_a, s.localAnchorA); put3(d.local_anchor_b, s.localAnchorB); d.global_length = s.globalLength; }
void fromRef(const ball_constraint& s, mi_ball_constraint& d) { put3(d.local_anchor_a, s.localAnchorA); put3(d.local_anchor_b, s.localAnchorB); }
void fromRef(const fixed_constraint& s, mi_fixed_constraint& d) { put4(d.initial_inv_rotation_difference, s.initialInvRotationDifference); put3(d.local_anchor_a, s.localAnchorA); put3(d.local_anchor_b, s.localAnchorB); }
void fromRef(const hinge_constraint& s, mi_hinge_constraint& d)
{
	put3(d.local_anchor_a, s.localAnchorA); put3(d.local_anchor_b, s.localAnchorB); put3(d.local_hinge_axis_a, s.localHingeAxisA); put3(d.local_hinge_axis_b, s.localHingeAxisB);
	d.min_rotation_limit = s.minRotationLimit; d.max_rotation_limit = s.maxRotationLimit; d.max_motor_torque = s.maxMotorTorque;
	d.motor_type = (uint32)s.motorType; d.motor_velocity_or_target_angle = s.motorVelocity;
	put3(d.local_hinge_tangent_a, s.localHingeTangentA); put3(d.local_hinge_bitangent_a, s.localHingeBitangentA); put3(d.local_hinge_tangent_b, s.localHingeTangentB);
}
void fromRef(const cone_twist_constraint& s, mi_cone_twist_constraint& d)
{
	put3(d.local_anchor_a, s.localAnchorA); put3(d.local_anchor_b, s.localAnchorB); put3(d.local_limit_axis_a, s.localLimitAxisA); put3(d.local_limit_axis_b, s.localLimitAxisB);
	put3(d.local_limit_tangent_a, s.localLimitTangentA); put3(d.local_limit_bitangent_a, s.localLimitBitangentA); put3(d.local_limit_tangent_b, s.localLimitTangentB);
	d.swing_limit = s.swingLimit; d.twist_limit = s.twistLimit;
	d.swing_motor_type = (uint32)s.swingMotorType; d.swing_motor_velocity_or_target_angle = s.swingMotorVelocity; d.max_swing_motor_torque = s.maxSwingMotorTorque;
	d.swing_motor_axis = s.swingMotorAxis; d.twist_motor_type = (uint32)s.twistMotorType; d.twist_motor_velocity_or_target_angle = s.twistMotorVelocity;
	d.max_twist_motor_torque = s.maxTwistMotorTorque;
}
void fromRef(const slider_constraint& s, mi_slider_constraint& d)
{
	put4(d.initial_inv_rotation_difference, s.initialInvRotationDifference); put3(d.local_anchor_a, s.localAnchorA); put3(d.local_anchor_b, s.localAnchorB);
	put3(d.local_axis_a, s.localAxisA); d.neg_distance_limit = s.negDistanceLimit; d.pos_distance_limit = s.posDistanceLimit;
	d.max_motor_force = s.maxMotorForce; d.motor_type = (uint32)s.motorType; d.motor_velocity_or_target_distance = s.motorVelocity;
}

template <typename mi_t, typename ref_t, typename handle_t>
int createConstraint(ref_world* w, uint32 type, uint32 ea, uint32 eb, const void* pod, uint32 bytes, uint32* out)
{
	if (bytes != sizeof(mi_t)) { return MI_ERR_INVALID_ARGUMENT; }
	ref_t c{};
	toRef(*(const mi_t*)pod, c);
	handle_t h = addConstraint(w->entities[ea], w->entities[eb], c);
	if (out) { *out = (uint32)w->constraintEntities[type].size(); }
	w->constraintEntities[type].push_back(h.entity);
	return MI_OK;
}

template <typename mi_t, typename ref_t, typename handle_t>
int accessConstraint(ref_world* w, uint32 type, uint32 id, void* outPod, const void* inPod, uint32 bytes)
{
	if (bytes != sizeof(mi_t) || id >= w->constraintEntities[type].size() || w->constraintEntities[type][id] == entt::null) { return MI_ERR_INVALID_ARGUMENT; }
	ref_t& c = getConstraint(w->scene, handle_t{ w->constraintEntities[type][id] });
	if (inPod) { toRef(*(const mi_t*)inPod, c); }
	if (outPod) { fromRef(c, *(mi_t*)outPod); }
	return MI_OK;
}

} // namespace

// ---------------------------------------------------------------- instrumentation taps (called from the patched physicsStepInternal)
void ref_tap_broadphase(const collider_pair* pairs, uint32 numPairs)
{
	if (ref_world* w = g_stepping) { w->bpPairs.assign(pairs, pairs + numPairs); }
}

void ref_tap_step(uint32 numRigidBodies, uint32 numColliders, const bounding_box* aabbs, const collider_union* worldSpaceColliders,
	uint32 numBroadphaseOverlaps, uint32 numCollisions, uint32 numContacts, const collision_contact* contacts, const constraint_body_pair* bodyPairs,
	const collider_pair* collidingPairs, const uint8* contactCountPerCollision)
{
	ref_world* w = g_stepping;
	if (!w) { return; }
	w->counts.num_rigid_bodies = numRigidBodies; w->counts.num_colliders = numColliders; w->counts.num_broadphase_overlaps = numBroadphaseOverlaps;
	w->counts.num_collisions = numCollisions; w->counts.num_contacts = numContacts; w->counts.num_colors = 0;
	w->aabbs.assign(aabbs, aabbs + numColliders);
	w->contacts.resize(numContacts);
	uint32 ci = 0;
	for (uint32 m = 0; m < numCollisions; ++m)
	{
		for (uint32 k = 0; k < contactCountPerCollision[m] && ci < numContacts; ++k, ++ci)
		{
			const collision_contact& c = contacts[ci]; mi_contact& o = w->contacts[ci];
			put3(o.point, c.point); o.penetration_depth = c.penetrationDepth; put3(o.normal, c.normal); o.friction_restitution = c.friction_restitution;
			o.collider_a = collidingPairs[m].colliderA; o.collider_b = collidingPairs[m].colliderB;
			o.body_a = bodyPairs[ci].rbA; o.body_b = bodyPairs[ci].rbB;
		}
	}
	(void)worldSpaceColliders;
}

extern "C" {

#define REF_API __attribute__((visibility("default")))

static void forgetDeadConstraints(ref_world* w);
REF_API int ref_version() { return 3; }
REF_API const char* ref_last_error() { return ""; }

// flags bit 0: physics_settings::simdBroadPhase / simdNarrowPhase / simdConstraintSolver (the reference's AVX2 path) on
REF_API int ref_world_create(int flags, ref_world** out)
{
	ref_world* w = new ref_world();
	w->arena.initialize(0, GB(32));
	bool simd = (flags & 1) != 0;
	w->settings.simdBroadPhase = simd; w->settings.simdNarrowPhase = simd; w->settings.simdConstraintSolver = simd;
	*out = w;
	return MI_OK;
}
REF_API void ref_world_destroy(ref_world* w) { delete w; }
// physics_settings::simdBroadPhase / simdNarrowPhase / simdConstraintSolver (src/physics/physics.h:394-396) can be flipped between steps
REF_API int ref_world_set_simd(ref_world* w, uint32 enable)
{
	w->settings.simdBroadPhase = w->settings.simdNarrowPhase = w->settings.simdConstraintSolver = enable != 0;
	return MI_OK;
}

REF_API int ref_entities_create(ref_world* w, uint32 count, const mi_entity_desc* descs, uint32* outFirst)
{
	if (outFirst) { *outFirst = (uint32)w->entities.size(); }
	for (uint32 i = 0; i < count; ++i)
	{
		const mi_entity_desc& d = descs[i];
		scene_entity e = w->scene.createEntity("e").addComponent<transform_component>(v3(d.position), q4(d.rotation));
		if (d.kind == MI_ENTITY_DYNAMIC || d.kind == MI_ENTITY_KINEMATIC)
		{
			e.addComponent<rigid_body_component>(d.kind == MI_ENTITY_KINEMATIC, d.gravity_factor, d.linear_damping, d.angular_damping);
			rigid_body_component& rb = e.getComponent<rigid_body_component>();
			rb.linearVelocity = v3(d.linear_velocity); rb.angularVelocity = v3(d.angular_velocity);
		}
		else if (d.kind == MI_ENTITY_TRIGGER)
		{
			ref_world* world = w;
			e.addComponent<trigger_component>(trigger_component{ [world](trigger_event ev)
			{
				mi_event o{}; o.type = ev.type == trigger_event_enter ? MI_EVENT_TRIGGER_ENTER : MI_EVENT_TRIGGER_LEAVE;
				o.entity_a = world->entityIdOfHandle[(uint32)ev.trigger.handle]; o.entity_b = world->entityIdOfHandle[(uint32)ev.other.handle];
				o.collider_a = o.collider_b = 0xFFFFFFFFu;
				world->events.push_back(o);
			} });
		}
		else if (d.kind == MI_ENTITY_FORCE_FIELD)
		{
			e.addComponent<force_field_component>(force_field_component{ vec3(0.f) });
		}
		w->entityIdOfHandle[(uint32)e.handle] = (uint32)w->entities.size();
		w->entities.push_back(e);
		w->kinds.push_back(d.kind);
	}
	return MI_OK;
}
REF_API int ref_entity_create(ref_world* w, const mi_entity_desc* d, uint32* out) { return ref_entities_create(w, 1, d, out); }
// game_scene::deleteEntity (src/scene/scene.cpp:124-150)
REF_API int ref_entity_destroy(ref_world* w, uint32 entity)
{
	if (entity >= w->entities.size() || w->kinds[entity] == MI_ENTITY_DESTROYED) { return MI_ERR_INVALID_ARGUMENT; }
	w->scene.deleteEntity(w->entities[entity]);
	w->kinds[entity] = MI_ENTITY_DESTROYED;
	forgetDeadConstraints(w);
	return MI_OK;
}

REF_API int ref_colliders_add(ref_world* w, uint32 count, const uint32* entities, const mi_collider_desc* descs)
{
	for (uint32 i = 0; i < count; ++i)
	{
		if (entities[i] >= w->entities.size()) { return MI_ERR_INVALID_ARGUMENT; }
		scene_entity& e = w->entities[entities[i]];
		e.addComponent<collider_component>(colliderFromDesc(w, descs[i]));
		entity_handle child = e.getComponent<physics_reference_component>().firstColliderEntity;
		w->colliderIdOfHandle[(uint32)child] = (uint32)w->colliderEntities.size();
		w->colliderEntities.push_back(child);
	}
	return MI_OK;
}
REF_API int ref_collider_add(ref_world* w, uint32 entity, const mi_collider_desc* d, uint32* out)
{
	if (out) { *out = (uint32)w->colliderEntities.size(); }
	return ref_colliders_add(w, 1, &entity, d);
}

REF_API int ref_hull_geometry_create(ref_world* w, const float* v, uint32 nv, const uint32* t, uint32 nt, uint32* out)
{
	std::vector<vec3> verts(nv);
	for (uint32 i = 0; i < nv; ++i) { verts[i] = v3(v + 3 * i); }
	std::vector<indexed_triangle16> tris(nt);
	for (uint32 i = 0; i < nt; ++i) { tris[i] = { (uint16)t[3 * i], (uint16)t[3 * i + 1], (uint16)t[3 * i + 2] }; }
	*out = (uint32)w->hullIds.size();
	w->hullIds.push_back(ref_allocate_hull_geometry(verts.data(), nv, tris.data(), nt));
	return MI_OK;
}

REF_API int ref_constraint_create(ref_world* w, uint32 type, uint32 ea, uint32 eb, const void* pod, uint32 bytes, uint32* out)
{
	if (ea >= w->entities.size() || eb >= w->entities.size()) { return MI_ERR_INVALID_ARGUMENT; }
	switch (type)
	{
		case MI_CONSTRAINT_DISTANCE: return createConstraint<mi_distance_constraint, distance_constraint, distance_constraint_handle>(w, type, ea, eb, pod, bytes, out);
		case MI_CONSTRAINT_BALL: return createConstraint<mi_ball_constraint, ball_constraint, ball_constraint_handle>(w, type, ea, eb, pod, bytes, out);
		case MI_CONSTRAINT_FIXED: return createConstraint<mi_fixed_constraint, fixed_constraint, fixed_constraint_handle>(w, type, ea, eb, pod, bytes, out);
		case MI_CONSTRAINT_HINGE: return createConstraint<mi_hinge_constraint, hinge_constraint, hinge_constraint_handle>(w, type, ea, eb, pod, bytes, out);
		case MI_CONSTRAINT_CONE_TWIST: return createConstraint<mi_cone_twist_constraint, cone_twist_constraint, cone_twist_constraint_handle>(w, type, ea, eb, pod, bytes, out);
		case MI_CONSTRAINT_SLIDER: return createConstraint<mi_slider_constraint, slider_constraint, slider_constraint_handle>(w, type, ea, eb, pod, bytes, out);
	}
	return MI_ERR_INVALID_ARGUMENT;
}

static int constraintAccess(ref_world* w, uint32 type, uint32 id, void* outPod, const void* inPod, uint32 bytes)
{
	switch (type)
	{
		case MI_CONSTRAINT_DISTANCE: return accessConstraint<mi_distance_constraint, distance_constraint, distance_constraint_handle>(w, type, id, outPod, inPod, bytes);
		case MI_CONSTRAINT_BALL: return accessConstraint<mi_ball_constraint, ball_constraint, ball_constraint_handle>(w, type, id, outPod, inPod, bytes);
		case MI_CONSTRAINT_FIXED: return accessConstraint<mi_fixed_constraint, fixed_constraint, fixed_constraint_handle>(w, type, id, outPod, inPod, bytes);
		case MI_CONSTRAINT_HINGE: return accessConstraint<mi_hinge_constraint, hinge_constraint, hinge_constraint_handle>(w, type, id, outPod, inPod, bytes);
		case MI_CONSTRAINT_CONE_TWIST: return accessConstraint<mi_cone_twist_constraint, cone_twist_constraint, cone_twist_constraint_handle>(w, type, id, outPod, inPod, bytes);
		case MI_CONSTRAINT_SLIDER: return accessConstraint<mi_slider_constraint, slider_constraint, slider_constraint_handle>(w, type, id, outPod, inPod, bytes);
	}
	return MI_ERR_INVALID_ARGUMENT;
}
REF_API int ref_constraint_update(ref_world* w, uint32 type, uint32 id, const void* pod, uint32 bytes) { return constraintAccess(w, type, id, nullptr, pod, bytes); }
REF_API int ref_constraint_get(ref_world* w, uint32 type, uint32 id, void* pod, uint32 bytes) { return constraintAccess(w, type, id, pod, nullptr, bytes); }
REF_API int ref_constraints_update(ref_world* w, uint32 type, uint32 count, const uint32* ids, const void* pods, uint32 podBytes)
{
	for (uint32 i = 0; i < count; ++i)
	{
		int rc = constraintAccess(w, type, ids[i], nullptr, (const char*)pods + (size_t)i * podBytes, podBytes);
		if (rc != MI_OK) { return rc; }
	}
	return MI_OK;
}

REF_API int ref_constraint_destroy(ref_world* w, uint32 type, uint32 id)
{
	if (type >= MI_CONSTRAINT_TYPE_COUNT || id >= w->constraintEntities[type].size() || w->constraintEntities[type][id] == entt::null) { return MI_ERR_INVALID_ARGUMENT; }
	entity_handle h = w->constraintEntities[type][id];
	switch (type)
	{
		case MI_CONSTRAINT_DISTANCE: deleteConstraint(w->scene, distance_constraint_handle{ h }); break;
		case MI_CONSTRAINT_BALL: deleteConstraint(w->scene, ball_constraint_handle{ h }); break;
		case MI_CONSTRAINT_FIXED: deleteConstraint(w->scene, fixed_constraint_handle{ h }); break;
		case MI_CONSTRAINT_HINGE: deleteConstraint(w->scene, hinge_constraint_handle{ h }); break;
		case MI_CONSTRAINT_CONE_TWIST: deleteConstraint(w->scene, cone_twist_constraint_handle{ h }); break;
		case MI_CONSTRAINT_SLIDER: deleteConstraint(w->scene, slider_constraint_handle{ h }); break;
	}
	w->constraintEntities[type][id] = entt::null;
	return MI_OK;
}
static void forgetDeadConstraints(ref_world* w)
{
	for (auto& list : w->constraintEntities) { for (auto& h : list) { if (!(h == entt::null) && !w->scene.registry.valid(h)) { h = entt::null; } } }
}
REF_API int ref_constraints_destroy_all(ref_world* w) { deleteAllConstraints(w->scene); forgetDeadConstraints(w); return MI_OK; }
REF_API int ref_entity_destroy_constraints(ref_world* w, uint32 entity)
{
	if (entity >= w->entities.size()) { return MI_ERR_INVALID_ARGUMENT; }
	deleteAllConstraintsFromEntity(w->entities[entity]);
	forgetDeadConstraints(w);
	return MI_OK;
}

REF_API int ref_constraint_create_from_global(ref_world* w, uint32 type, uint32 ea, uint32 eb, const float* anchor, const float* axis, float l0, float l1, uint32* out)
{
	if (ea >= w->entities.size() || eb >= w->entities.size()) { return MI_ERR_INVALID_ARGUMENT; }
	scene_entity& a = w->entities[ea]; scene_entity& b = w->entities[eb];
	entity_handle h = entt::null;
	switch (type)
	{
		// the ABI passes the second global anchor of a distance constraint in `axis` (include/mi_physics.h)
		case MI_CONSTRAINT_DISTANCE: h = addDistanceConstraintFromGlobalPoints(a, b, v3(anchor), v3(axis)).entity; break;
		case MI_CONSTRAINT_BALL: h = addBallConstraintFromGlobalPoints(a, b, v3(anchor)).entity; break;
		case MI_CONSTRAINT_FIXED: h = addFixedConstraintFromGlobalPoints(a, b, v3(anchor)).entity; break;
		case MI_CONSTRAINT_HINGE: h = addHingeConstraintFromGlobalPoints(a, b, v3(anchor), v3(axis), l0, l1).entity; break;
		case MI_CONSTRAINT_CONE_TWIST: h = addConeTwistConstraintFromGlobalPoints(a, b, v3(anchor), v3(axis), l0, l1).entity; break;
		case MI_CONSTRAINT_SLIDER: h = addSliderConstraintFromGlobalPoints(a, b, v3(anchor), v3(axis), l0, l1).entity; break;
		default: return MI_ERR_INVALID_ARGUMENT;
	}
	if (out) { *out = (uint32)w->constraintEntities[type].size(); }
	w->constraintEntities[type].push_back(h);
	return MI_OK;
}

REF_API int ref_entity_apply_force(ref_world* w, uint32 entity, const float* f, const float* t)
{
	if (entity >= w->entities.size() || !w->entities[entity].hasComponent<rigid_body_component>()) { return MI_ERR_INVALID_ARGUMENT; }
	rigid_body_component& rb = w->entities[entity].getComponent<rigid_body_component>();
	if (f) { rb.forceAccumulator += v3(f); }
	if (t) { rb.torqueAccumulator += v3(t); }
	return MI_OK;
}
REF_API int ref_entities_apply_forces(ref_world* w, uint32 count, const uint32* ents, const float* f, const float* t)
{
	for (uint32 i = 0; i < count; ++i)
	{
		int rc = ref_entity_apply_force(w, ents[i], f ? f + 3 * i : nullptr, t ? t + 3 * i : nullptr);
		if (rc != MI_OK) { return rc; }
	}
	return MI_OK;
}
REF_API int ref_entity_set_force(ref_world* w, uint32 entity, const float* f)
{
	if (entity >= w->entities.size() || !w->entities[entity].hasComponent<force_field_component>()) { return MI_ERR_INVALID_ARGUMENT; }
	w->entities[entity].getComponent<force_field_component>().force = v3(f);
	return MI_OK;
}
REF_API int ref_world_test_interactions(ref_world* w, uint32 count, const float* origins, const float* directions, const float* strengths, const uint32* ranges)
{
	if (ranges) { return MI_ERR_UNSUPPORTED; }      // entity ranges are an addition of the product ABI
	for (uint32 i = 0; i < count; ++i) { testPhysicsInteraction(w->scene, ray{ v3(origins + 3 * i), v3(directions + 3 * i) }, strengths ? strengths[i] : 1000.f); }
	return MI_OK;
}

// --- heightmap terrain
REF_API int ref_heightmap_create(ref_world* w, uint32 chunksPerDim, float chunkSize, float restitution, float friction)
{
	if (w->heightmapEntity) { return MI_ERR_INVALID_ARGUMENT; }
	w->heightmapEntity = w->scene.createEntity("terrain");
	w->heightmapEntity.addComponent<heightmap_collider_component>(chunksPerDim, chunkSize, physics_material{ physics_material_type_none, restitution, friction, 0.f });
	return MI_OK;
}
REF_API int ref_heightmap_set_chunk_heights(ref_world* w, uint32 x, uint32 z, const uint16* heights)
{
	if (!w->heightmapEntity) { return MI_ERR_INVALID_ARGUMENT; }
	// the reference's chunk keeps the caller's pointer (heightmap_collider.cpp setHeights): give it memory that lives as long as the world
	static std::vector<std::unique_ptr<uint16[]>> keep;
	const uint32 n = TERRAIN_LOD_0_VERTICES_PER_DIMENSION * TERRAIN_LOD_0_VERTICES_PER_DIMENSION;
	keep.emplace_back(new uint16[n]);
	memcpy(keep.back().get(), heights, n * sizeof(uint16));
	w->heightmapEntity.getComponent<heightmap_collider_component>().collider(x, z).setHeights(keep.back().get());
	return MI_OK;
}
REF_API int ref_heightmap_update(ref_world* w, const float* minCorner, float amplitudeScale)
{
	if (!w->heightmapEntity) { return MI_ERR_INVALID_ARGUMENT; }
	w->heightmapEntity.getComponent<heightmap_collider_component>().update(v3(minCorner), amplitudeScale);
	return MI_OK;
}
REF_API int ref_heightmap_get_height(ref_world* w, float x, float z, float* out)
{
	if (!w->heightmapEntity) { return MI_ERR_INVALID_ARGUMENT; }
	*out = w->heightmapEntity.getComponent<heightmap_collider_component>().getHeightAt(vec2(x, z));
	return MI_OK;
}

// --- events (collisionBeginCallback / collisionEndCallback of physics_settings)
REF_API int ref_world_enable_events(ref_world* w, uint32 enable)
{
	w->eventsEnabled = enable != 0;
	w->events.clear();
	if (!enable) { w->settings.collisionBeginCallback = nullptr; w->settings.collisionEndCallback = nullptr; return MI_OK; }
	w->settings.collisionBeginCallback = [w](const collision_begin_event& ev)
	{
		mi_event o{}; o.type = MI_EVENT_COLLISION_BEGIN;
		o.entity_a = w->entityIdOfHandle[(uint32)ev.entityA.handle]; o.entity_b = w->entityIdOfHandle[(uint32)ev.entityB.handle];
		o.collider_a = w->colliderIdOfHandle[(uint32)entt::to_entity(w->scene.registry, ev.colliderA)];
		o.collider_b = w->colliderIdOfHandle[(uint32)entt::to_entity(w->scene.registry, ev.colliderB)];
		put3(o.point, ev.position); put3(o.normal, ev.normal); put3(o.relative_velocity, ev.relativeVelocity);
		w->events.push_back(o);
	};
	w->settings.collisionEndCallback = [w](const collision_end_event& ev)
	{
		mi_event o{}; o.type = MI_EVENT_COLLISION_END;
		o.entity_a = w->entityIdOfHandle[(uint32)ev.entityA.handle]; o.entity_b = w->entityIdOfHandle[(uint32)ev.entityB.handle];
		o.collider_a = w->colliderIdOfHandle[(uint32)entt::to_entity(w->scene.registry, ev.colliderA)];
		o.collider_b = w->colliderIdOfHandle[(uint32)entt::to_entity(w->scene.registry, ev.colliderB)];
		w->events.push_back(o);
	};
	return MI_OK;
}
REF_API int ref_world_poll_events(ref_world* w, mi_event* out, uint32 cap, uint32* count)
{
	*count = (uint32)w->events.size();
	if (!out) { return MI_OK; }
	if (cap < w->events.size()) { return MI_ERR_CAPACITY; }
	memcpy(out, w->events.data(), w->events.size() * sizeof(mi_event));
	w->events.clear();
	return MI_OK;
}

// --- stepping
static void applySettings(ref_world* w, const mi_step_settings* s)
{
	w->settings.fixedFrameRate = s->fixed_frame_rate != 0; w->settings.frameRate = s->frame_rate;
	w->settings.maxPhysicsIterationsPerFrame = s->max_physics_iterations_per_frame; w->settings.numRigidSolverIterations = s->num_rigid_solver_iterations;
}
REF_API int ref_world_step(ref_world* w, const mi_step_settings* s, float dt)
{
	applySettings(w, s);
	g_stepping = w;
	uint32 axis = ref_sap_sorting_axis(w->scene);     // the axis this step's sweep sorts along (mi_step_counts::sorting_axis); exact for one internal step per call
	physicsStep(w->scene, w->arena, w->timer, w->settings, dt);
	g_stepping = nullptr;
	w->counts.sorting_axis = axis;
	return MI_OK;
}
REF_API int ref_world_step_fixed(ref_world* w, const mi_step_settings* s, float dt, uint32 n)
{
	applySettings(w, s);
	w->settings.fixedFrameRate = false;       // physicsStep then runs exactly one physicsStepInternal(dt) and copies physics_transform1 to the transform
	g_stepping = w;
	uint32 axis = 0;
	for (uint32 i = 0; i < n; ++i) { axis = ref_sap_sorting_axis(w->scene); physicsStep(w->scene, w->arena, w->timer, w->settings, dt); }
	g_stepping = nullptr;
	w->counts.sorting_axis = axis;
	return MI_OK;
}
REF_API int ref_world_set_cloth_iterations(ref_world* w, uint32 v, uint32 p, uint32 d)
{
	w->settings.numClothVelocityIterations = v; w->settings.numClothPositionIterations = p; w->settings.numClothDriftIterations = d;
	return MI_OK;
}

// --- cloth (cloth_component, src/physics/cloth.h)
REF_API int ref_cloth_create(ref_world* w, const mi_cloth_desc* d, uint32* out)
{
	scene_entity e = w->scene.createEntity("cloth");
	e.addComponent<cloth_component>(d->width, d->height, d->grid_size_x, d->grid_size_y, d->total_mass, d->stiffness, d->damping, d->gravity_factor);
	if (out) { *out = (uint32)w->cloths.size(); }
	w->cloths.push_back(e);
	return MI_OK;
}
REF_API int ref_cloth_set_fixed_vertices(ref_world* w, uint32 cloth, const float* p, const float* r, uint32 moveRigid)
{
	if (cloth >= w->cloths.size()) { return MI_ERR_INVALID_ARGUMENT; }
	w->cloths[cloth].getComponent<cloth_component>().setWorldPositionOfFixedVertices(trs(v3(p), q4(r)), moveRigid != 0);
	return MI_OK;
}
REF_API int ref_cloth_set_properties(ref_world* w, uint32 cloth, float totalMass, float stiffness, float damping, float gravityFactor)
{
	if (cloth >= w->cloths.size()) { return MI_ERR_INVALID_ARGUMENT; }
	cloth_component& c = w->cloths[cloth].getComponent<cloth_component>();
	c.totalMass = totalMass; c.stiffness = stiffness; c.damping = damping; c.gravityFactor = gravityFactor;
	return MI_OK;
}
REF_API int ref_cloth_get_state(ref_world* w, uint32 cloth, float* pos, float* vel, uint32 cap)
{
	if (cloth >= w->cloths.size()) { return MI_ERR_INVALID_ARGUMENT; }
	const cloth_component& c = w->cloths[cloth].getComponent<cloth_component>();
	if (cap < c.positions.size()) { return MI_ERR_CAPACITY; }
	for (size_t i = 0; i < c.positions.size(); ++i)
	{
		if (pos) { put3(pos + 3 * i, c.positions[i]); }
		if (vel) { put3(vel + 3 * i, c.velocities[i]); }
	}
	return MI_OK;
}

// --- binary entity stream (serializeEntityToMemory / deserializeEntityFromMemory, src/scene/serialization_binary.cpp:484-497).
// That translation unit needs the renderer's component types and cannot be compiled here; what follows walks the SAME component
// list in the SAME order (serialized_components, lines 105-133) and writes each physics component exactly as its
// serializeToMemoryStream does — `stream.write(component)` = the raw struct image, taken here from the reference's own struct
// definitions — with `false` for every renderer / terrain-rendering component.  Entity handles inside the stream are this ABI's
// entity ids (the reference writes EnTT identifiers, which only mean something inside one registry).
extern "C++" {
namespace {
struct byte_writer
{
	uint8* buffer; uint64 cap; uint64 offset = 0; bool overflow = false;
	template <typename T> void write(const T& t) { if (offset + sizeof(T) > cap) { overflow = true; offset += sizeof(T); return; } memcpy(buffer + offset, &t, sizeof(T)); offset += sizeof(T); }
};
}
}
static_assert(sizeof(tag_component) == 16 && sizeof(transform_component) == 48 && sizeof(rigid_body_component) == 112 && sizeof(force_field_component) == 12, "component images");
static_assert(sizeof(collider_union) == 80 && sizeof(physics_material) == 16 && sizeof(constraint_type) == 4, "collider image");
static_assert(sizeof(distance_constraint) == 28 && sizeof(ball_constraint) == 24 && sizeof(fixed_constraint) == 48 && sizeof(hinge_constraint) == 104 &&
	sizeof(cone_twist_constraint) == 120 && sizeof(slider_constraint) == 80, "constraint images");
REF_API int ref_entity_serialize(ref_world* w, uint32 entity, void* out, uint64 cap, uint64* outSize)
{
	if (entity >= w->entities.size() || w->kinds[entity] == MI_ENTITY_DESTROYED) { return MI_ERR_INVALID_ARGUMENT; }
	scene_entity e = w->entities[entity];
	byte_writer s{ (uint8*)out, out ? cap : 0 };
	auto flag = [&](bool has) { s.write(has); return has; };
	if (flag(e.hasComponent<tag_component>())) { s.write(e.getComponent<tag_component>()); }
	if (flag(e.hasComponent<transform_component>())) { s.write(e.getComponent<transform_component>()); }
	flag(false); flag(false); flag(false);                              // position / position_rotation / position_scale
	flag(e.hasComponent<dynamic_transform_component>());               // no payload
	flag(false); flag(false); flag(false);                              // mesh, point light, spot light
	if (flag(e.hasComponent<rigid_body_component>())) { s.write(e.getComponent<rigid_body_component>()); }
	if (flag(e.hasComponent<force_field_component>())) { s.write(e.getComponent<force_field_component>()); }
	if (flag(e.hasComponent<cloth_component>()))
	{
		const cloth_component& c = e.getComponent<cloth_component>();
		s.write(c.width); s.write(c.height); s.write(c.gridSizeX); s.write(c.gridSizeY); s.write(c.totalMass); s.write(c.stiffness); s.write(c.damping); s.write(c.gravityFactor);
	}
	flag(false);                                                        // cloth_render_component
	if (flag(e.hasComponent<physics_reference_component>()))
	{
		const physics_reference_component& ref = e.getComponent<physics_reference_component>();
		s.write(ref.numColliders);
		for (collider_component& collider : collider_component_iterator(e)) { collider_union u; memcpy(&u, &collider, sizeof(u)); if (u.type == collider_type_hull) { u.hull.geometryPtr = nullptr; for (uint32 k = 0; k < w->hullIds.size(); ++k) { if (w->hullIds[k] == collider.hull.geometryIndex) { u.hull.geometryIndex = k; } } } s.write(u); }
		s.write(ref.numConstraints);
		for (auto [constraintEntity, constraintType] : constraint_entity_iterator(e))
		{
			auto& cr = constraintEntity.getComponent<constraint_entity_reference_component>();
			s.write(constraintType);
			s.write((uint32)w->entityIdOfHandle[(uint32)cr.entityA]); s.write((uint32)w->entityIdOfHandle[(uint32)cr.entityB]);
			switch (constraintType)
			{
				case constraint_type_distance: s.write(constraintEntity.getComponent<distance_constraint>()); break;
				case constraint_type_ball: s.write(constraintEntity.getComponent<ball_constraint>()); break;
				case constraint_type_fixed: s.write(constraintEntity.getComponent<fixed_constraint>()); break;
				case constraint_type_hinge: s.write(constraintEntity.getComponent<hinge_constraint>()); break;
				case constraint_type_cone_twist: s.write(constraintEntity.getComponent<cone_twist_constraint>()); break;
				case constraint_type_slider: s.write(constraintEntity.getComponent<slider_constraint>()); break;
				default: break;
			}
		}
	}
	flag(false);                                                        // terrain_component
	if (flag(e.hasComponent<heightmap_collider_component>()))
	{
		const heightmap_collider_component& h = e.getComponent<heightmap_collider_component>();
		s.write(h.chunksPerDim); s.write(h.chunkSize); s.write(h.material);
	}
	flag(false); flag(false); flag(false);                              // grass, proc placement, water
	*outSize = s.offset;
	return s.overflow ? MI_ERR_CAPACITY : MI_OK;
}

// --- read-back
REF_API int ref_world_num_entities(ref_world* w, uint32* out) { *out = (uint32)w->entities.size(); return MI_OK; }
static int getTransforms(ref_world* w, float* p, float* r, uint32 cap, bool physics)
{
	uint32 n = (uint32)w->entities.size();
	if (cap < n) { return MI_ERR_CAPACITY; }
	for (uint32 i = 0; i < n; ++i)
	{
		if (w->kinds[i] == MI_ENTITY_DESTROYED) { if (p) { put3(p + 3 * i, vec3(0.f)); } if (r) { put4(r + 4 * i, quat(0.f, 0.f, 0.f, 1.f)); } continue; }
		scene_entity& e = w->entities[i];
		const trs* t = &e.getComponent<transform_component>();
		if (physics) { if (auto* pt = e.getComponentIfExists<physics_transform1_component>()) { t = pt; } }
		if (p) { put3(p + 3 * i, t->position); }
		if (r) { put4(r + 4 * i, t->rotation); }
	}
	return MI_OK;
}
REF_API int ref_world_get_transforms(ref_world* w, float* p, float* r, uint32 cap) { return getTransforms(w, p, r, cap, false); }
REF_API int ref_world_get_physics_transforms(ref_world* w, float* p, float* r, uint32 cap) { return getTransforms(w, p, r, cap, true); }
REF_API int ref_world_get_velocities(ref_world* w, float* lin, float* ang, uint32 cap)
{
	uint32 n = (uint32)w->entities.size();
	if (cap < n) { return MI_ERR_CAPACITY; }
	for (uint32 i = 0; i < n; ++i)
	{
		vec3 v(0.f), a(0.f);
		if (w->kinds[i] == MI_ENTITY_DESTROYED) { if (lin) { put3(lin + 3 * i, v); } if (ang) { put3(ang + 3 * i, a); } continue; }
		if (auto* rb = w->entities[i].getComponentIfExists<rigid_body_component>()) { v = rb->linearVelocity; a = rb->angularVelocity; }
		if (lin) { put3(lin + 3 * i, v); }
		if (ang) { put3(ang + 3 * i, a); }
	}
	return MI_OK;
}
REF_API int ref_world_get_mass_properties(ref_world* w, float* invMass, float* invInertia, float* cog, uint32 cap)
{
	uint32 n = (uint32)w->entities.size();
	if (cap < n) { return MI_ERR_CAPACITY; }
	for (uint32 i = 0; i < n; ++i)
	{
		float im = 0.f; mat3 ii = mat3::zero; vec3 c(0.f);
		if (w->kinds[i] == MI_ENTITY_DESTROYED) { if (invMass) { invMass[i] = im; } if (invInertia) { memcpy(invInertia + 9 * i, &ii, 36); } if (cog) { put3(cog + 3 * i, c); } continue; }
		if (auto* rb = w->entities[i].getComponentIfExists<rigid_body_component>()) { im = rb->invMass; ii = rb->invInertia; c = rb->localCOGPosition; }
		if (invMass) { invMass[i] = im; }
		if (invInertia) { memcpy(invInertia + 9 * i, &ii, 36); }
		if (cog) { put3(cog + 3 * i, c); }
	}
	return MI_OK;
}
REF_API int ref_world_get_counts(ref_world* w, mi_step_counts* out) { *out = w->counts; return MI_OK; }
REF_API int ref_world_get_contacts(ref_world* w, mi_contact* out, uint32 cap, uint32* count)
{
	*count = (uint32)w->contacts.size();
	if (!out) { return MI_OK; }
	if (cap < w->contacts.size()) { return MI_ERR_CAPACITY; }
	memcpy(out, w->contacts.data(), w->contacts.size() * sizeof(mi_contact));
	return MI_OK;
}
REF_API int ref_world_get_aabbs(ref_world* w, float* out6, uint32 cap)
{
	if (cap < w->aabbs.size()) { return MI_ERR_CAPACITY; }
	for (size_t i = 0; i < w->aabbs.size(); ++i) { put3(out6 + 6 * i, w->aabbs[i].minCorner); put3(out6 + 6 * i + 3, w->aabbs[i].maxCorner); }
	return MI_OK;
}
REF_API int ref_world_get_broadphase_pairs(ref_world* w, uint32* out2, uint32 cap, uint32* count)
{
	*count = (uint32)w->bpPairs.size();
	if (!out2) { return MI_OK; }
	if (cap < *count) { return MI_ERR_CAPACITY; }
	for (size_t i = 0; i < w->bpPairs.size(); ++i) { out2[2 * i] = w->bpPairs[i].colliderA; out2[2 * i + 1] = w->bpPairs[i].colliderB; }
	return MI_OK;
}
REF_API int ref_world_get_body_states(ref_world* w, uint32 n, const uint32* ents, float* out)
{
	for (uint32 i = 0; i < n; ++i)
	{
		if (ents[i] >= w->entities.size() || !w->entities[ents[i]].hasComponent<rigid_body_component>()) { return MI_ERR_INVALID_ARGUMENT; }
		scene_entity& e = w->entities[ents[i]];
		const rigid_body_component& rb = e.getComponent<rigid_body_component>(); const trs& t = e.getComponent<physics_transform1_component>();
		float* o = out + 13 * (size_t)i;
		put3(o, t.position); put4(o + 3, t.rotation); put3(o + 7, rb.linearVelocity); put3(o + 10, rb.angularVelocity);
	}
	return MI_OK;
}
REF_API int ref_world_set_body_states(ref_world* w, uint32 n, const uint32* ents, const float* in)
{
	for (uint32 i = 0; i < n; ++i)
	{
		if (ents[i] >= w->entities.size() || !w->entities[ents[i]].hasComponent<rigid_body_component>()) { return MI_ERR_INVALID_ARGUMENT; }
		scene_entity& e = w->entities[ents[i]];
		rigid_body_component& rb = e.getComponent<rigid_body_component>(); trs& t = e.getComponent<physics_transform1_component>();
		const float* s = in + 13 * (size_t)i;
		t.position = v3(s); t.rotation = q4(s + 3); rb.linearVelocity = v3(s + 7); rb.angularVelocity = v3(s + 10);
	}
	return MI_OK;
}

} // extern "C"
