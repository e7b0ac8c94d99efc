// physics_mi355x.cpp — the file a maintainer of pkurth/D3D12Renderer adds under src/physics/ to run the rigid-body step on libmi_physics.so
// (INTEGRATION.md §2).  It is compiled here, against the reference's REAL headers (src/physics/physics.h, src/scene/scene.h), by
// oracle/refbuild/binding/build_binding.py — test infrastructure like the rest of oracle/refbuild: nothing of the reference is committed, this file
// is ours.  tests/test_gpu_binding.py runs the result on the GPU.
//
// What it replaces: the body of physicsStep (src/physics/physics.cpp:1364-1413).  What it needs from the scene: nothing but the components the
// reference's own step reads — it mirrors the ECS into an mi_world whenever the physics topology changed and writes the poses and velocities back
// after every step, so game code keeps reading / writing transform_component and rigid_body_component as before:
//   scene_entity::addComponent<collider_component | rigid_body_component | transform_component | ...>   (src/scene/scene.h:35-112)  -> miOnPhysicsComponentChanged (hook, one line)
//   addXxxConstraint... / deleteConstraint / deleteEntity                                             (src/physics/physics.cpp:128-552) -> noticed by pool sizes
//   getConstraint(scene, handle).maxMotorTorque = ...  (mutable references: motors, limits)              -> PODs compared and re-sent (mi_constraints_update)
//   rb.forceAccumulator += f, rb.linearVelocity = v, transform teleports                                 -> compared with what was written back, pushed before the step
// Index conventions are the reference's own: body index = pool position of rigid_body_component, collider world index = reverse pool position
// (src/physics/physics.cpp:631-664), constraints of a type in pool order — the mirror creates everything in pool order, so contacts, events and
// debug dumps mean the same thing on both sides.
#include "physics/physics.h"
#include "physics/collision_broad.h"
#include "scene/scene.h"
#include "mi_physics.h"
#include "mi_constraints.h"
#include "physics_mi355x.h"

// the reference's constraint structs ARE the library's PODs in memory (fixed / slider: plus the tail padding of their leading quat)
static_assert(sizeof(distance_constraint) == MI_REF_SIZEOF_DISTANCE_CONSTRAINT && sizeof(ball_constraint) == MI_REF_SIZEOF_BALL_CONSTRAINT, "constraints.h:73-80, 129-135");
static_assert(sizeof(fixed_constraint) == MI_REF_SIZEOF_FIXED_CONSTRAINT && sizeof(hinge_constraint) == MI_REF_SIZEOF_HINGE_CONSTRAINT, "constraints.h:175-183, 229-257");
static_assert(sizeof(cone_twist_constraint) == MI_REF_SIZEOF_CONE_TWIST_CONSTRAINT && sizeof(slider_constraint) == MI_REF_SIZEOF_SLIDER_CONSTRAINT, "constraints.h:346-380, 497-520");
static_assert(offsetof(hinge_constraint, localHingeTangentA) == offsetof(mi_hinge_constraint, local_hinge_tangent_a) && offsetof(slider_constraint, motorType) == offsetof(mi_slider_constraint, motor_type) &&
              offsetof(cone_twist_constraint, maxTwistMotorTorque) == offsetof(mi_cone_twist_constraint, max_twist_motor_torque) && offsetof(fixed_constraint, localAnchorB) == offsetof(mi_fixed_constraint, local_anchor_b), "field for field");

namespace
{
	struct body_cache { vec3 position; quat rotation; vec3 linearVelocity, angularVelocity; };

	template <typename constraint_t> struct constraint_mirror { std::vector<constraint_t> sent; std::vector<uint32> ids; };

	struct mi_backend_context
	{
		mi_world* world = nullptr;
		bool dirty = true;
		uint32 counts[12] = {};                      // pool sizes at the last sync: a change is a topology change
		std::vector<entity_handle> entityOfIndex;    // mi entity id -> scene entity
		std::unordered_map<uint32, uint32> indexOfEntity;
		std::vector<uint32> bodyEntityIds;           // mi entity id of rigid body i (pool order)
		std::vector<body_cache> written;             // what the last step wrote back, per rigid body (pool order)
		constraint_mirror<distance_constraint> distance; constraint_mirror<ball_constraint> ball; constraint_mirror<fixed_constraint> fixed;
		constraint_mirror<hinge_constraint> hinge; constraint_mirror<cone_twist_constraint> coneTwist; constraint_mirror<slider_constraint> slider;
		std::string error;
		~mi_backend_context() { if (world) { mi_world_destroy(world); } }
	};

	bool check(mi_backend_context& ctx, int rc, const char* what)
	{
		if (rc == MI_OK) { return true; }
		ctx.error = std::string(what) + ": " + mi_last_error();
		std::cerr << "[physics_mi355x] " << ctx.error << '\n';
		return false;
	}

	template <typename component_t> uint32 poolSize(game_scene& scene) { return scene.numberOfComponentsOfType<component_t>(); }

	void currentCounts(game_scene& scene, uint32* c)
	{
		c[0] = poolSize<rigid_body_component>(scene); c[1] = poolSize<collider_component>(scene); c[2] = poolSize<force_field_component>(scene); c[3] = poolSize<trigger_component>(scene);
		c[4] = poolSize<distance_constraint>(scene); c[5] = poolSize<ball_constraint>(scene); c[6] = poolSize<fixed_constraint>(scene); c[7] = poolSize<hinge_constraint>(scene);
		c[8] = poolSize<cone_twist_constraint>(scene); c[9] = poolSize<slider_constraint>(scene); c[10] = poolSize<cloth_component>(scene); c[11] = 0;
	}

	const trs& physicsPose(scene_entity entity)
	{
		// what getWorldSpaceColliders reads (src/physics/physics.cpp:642-644): physics_transform1 of a rigid body, else the transform, else identity
		if (physics_transform1_component* p = entity.getComponentIfExists<physics_transform1_component>()) { return *p; }
		if (transform_component* t = entity.getComponentIfExists<transform_component>()) { return *t; }
		return trs::identity;
	}

	uint32 mirrorEntity(mi_backend_context& ctx, scene_entity entity, uint32 kind)
	{
		auto it = ctx.indexOfEntity.find((uint32)entity.handle);
		if (it != ctx.indexOfEntity.end()) { return it->second; }
		mi_entity_desc d = {};
		const trs& t = physicsPose(entity);
		memcpy(d.position, &t.position, 12); memcpy(d.rotation, &t.rotation, 16);
		d.kind = kind; d.gravity_factor = 1.f;
		if (rigid_body_component* rb = entity.getComponentIfExists<rigid_body_component>())
		{
			memcpy(d.linear_velocity, &rb->linearVelocity, 12); memcpy(d.angular_velocity, &rb->angularVelocity, 12);
			d.gravity_factor = rb->gravityFactor; d.linear_damping = rb->linearDamping; d.angular_damping = rb->angularDamping;
			d.kind = (rb->invMass == 0.f) ? MI_ENTITY_KINEMATIC : MI_ENTITY_DYNAMIC;    // rigid_body_component(kinematic, ...) (src/physics/rigid_body.cpp:9-27)
		}
		uint32 id = 0;
		if (!check(ctx, mi_entity_create(ctx.world, &d, &id), "mi_entity_create")) { return 0; }
		if (force_field_component* ff = entity.getComponentIfExists<force_field_component>()) { check(ctx, mi_entity_set_force(ctx.world, id, ff->force.data), "mi_entity_set_force"); }
		ctx.indexOfEntity[(uint32)entity.handle] = id;
		if (ctx.entityOfIndex.size() <= id) { ctx.entityOfIndex.resize(id + 1, entt::null); }
		ctx.entityOfIndex[id] = entity.handle;
		return id;
	}

	mi_collider_desc colliderDesc(const collider_component& c)
	{
		// collider_union (src/physics/physics.h:84-106): the shape union copied field by field, the material 1:1
		mi_collider_desc d = {};
		d.type = (uint32)c.type; d.restitution = c.material.restitution; d.friction = c.material.friction; d.density = c.material.density;
		switch (c.type)
		{
			case collider_type_sphere: memcpy(d.shape, &c.sphere.center, 12); d.shape[3] = c.sphere.radius; break;
			case collider_type_capsule: memcpy(d.shape, &c.capsule.positionA, 12); memcpy(d.shape + 3, &c.capsule.positionB, 12); d.shape[6] = c.capsule.radius; break;
			case collider_type_cylinder: memcpy(d.shape, &c.cylinder.positionA, 12); memcpy(d.shape + 3, &c.cylinder.positionB, 12); d.shape[6] = c.cylinder.radius; break;
			case collider_type_aabb: memcpy(d.shape, &c.aabb.minCorner, 12); memcpy(d.shape + 3, &c.aabb.maxCorner, 12); break;
			case collider_type_obb: memcpy(d.shape, &c.obb.rotation, 16); memcpy(d.shape + 4, &c.obb.center, 12); memcpy(d.shape + 7, &c.obb.radius, 12); break;
			default: break;      // hulls: the geometry table is private to physics.cpp (boundingHullGeometries): handed over by mi_hull_geometry_create where it is filled
		}
		return d;
	}

	template <typename constraint_t>
	bool mirrorConstraints(mi_backend_context& ctx, game_scene& scene, uint32 type, constraint_mirror<constraint_t>& m)
	{
		const uint32 n = poolSize<constraint_t>(scene);
		m.sent.clear(); m.ids.clear();
		for (uint32 i = 0; i < n; ++i)          // pool order = the order the reference solves them in (src/physics/physics.cpp:792-804)
		{
			constraint_t& c = scene.getComponentAtIndex<constraint_t>(i);
			scene_entity ce = scene.getEntityFromComponentAtIndex<constraint_t>(i);
			const constraint_entity_reference_component& ref = ce.getComponent<constraint_entity_reference_component>();
			auto a = ctx.indexOfEntity.find((uint32)ref.entityA), b = ctx.indexOfEntity.find((uint32)ref.entityB);
			if (a == ctx.indexOfEntity.end() || b == ctx.indexOfEntity.end()) { ctx.error = "constraint between entities without a rigid body"; return false; }
			uint32 id = 0;
			// the reference's struct as it lies in memory: no member-wise conversion (include/mi_constraints.h "Reference layout")
			if (!check(ctx, mi_constraint_create(ctx.world, type, a->second, b->second, &c, sizeof(constraint_t), &id), "mi_constraint_create")) { return false; }
			m.sent.push_back(c); m.ids.push_back(id);
		}
		return true;
	}

	template <typename constraint_t>
	bool resendEditedConstraints(mi_backend_context& ctx, game_scene& scene, uint32 type, constraint_mirror<constraint_t>& m)
	{
		// getConstraint() hands out mutable references (motors, limits are edited live, e.g. by the ragdoll controller): re-send what changed
		for (uint32 i = 0; i < (uint32)m.sent.size(); ++i)
		{
			const constraint_t& c = scene.getComponentAtIndex<constraint_t>(i);
			if (memcmp(&c, &m.sent[i], sizeof(constraint_t)) != 0)
			{
				if (!check(ctx, mi_constraint_update(ctx.world, type, m.ids[i], &c, sizeof(constraint_t)), "mi_constraint_update")) { return false; }
				m.sent[i] = c;
			}
		}
		return true;
	}

	bool syncScene(mi_backend_context& ctx, game_scene& scene)
	{
		if (ctx.world) { mi_world_destroy(ctx.world); ctx.world = nullptr; }
		ctx.entityOfIndex.clear(); ctx.indexOfEntity.clear(); ctx.bodyEntityIds.clear(); ctx.written.clear(); ctx.error.clear();
		mi_world_desc wd = { 0, 0 };
		if (!check(ctx, mi_world_create(&wd, &ctx.world), "mi_world_create")) { return false; }
		// rigid bodies, force fields, triggers: in pool order, so that the library's dense indices are the reference's getComponentIndex values
		const uint32 nb = poolSize<rigid_body_component>(scene);
		for (uint32 i = 0; i < nb; ++i) { ctx.bodyEntityIds.push_back(mirrorEntity(ctx, scene.getEntityFromComponentAtIndex<rigid_body_component>(i), MI_ENTITY_DYNAMIC)); }
		for (uint32 i = 0; i < poolSize<force_field_component>(scene); ++i) { mirrorEntity(ctx, scene.getEntityFromComponentAtIndex<force_field_component>(i), MI_ENTITY_FORCE_FIELD); }
		for (uint32 i = 0; i < poolSize<trigger_component>(scene); ++i) { mirrorEntity(ctx, scene.getEntityFromComponentAtIndex<trigger_component>(i), MI_ENTITY_TRIGGER); }
		// colliders in pool order (world index = numColliders - 1 - pool position on both sides); an entity first seen here is a static collider's
		const uint32 nc = poolSize<collider_component>(scene);
		for (uint32 i = 0; i < nc; ++i)
		{
			const collider_component& c = scene.getComponentAtIndex<collider_component>(i);
			if (c.type == collider_type_hull) { ctx.error = "hull colliders: hand the geometry over with mi_hull_geometry_create where allocateBoundingHullGeometry fills it (src/physics/physics.cpp:58-84)"; return false; }
			const uint32 parent = mirrorEntity(ctx, scene_entity{ c.parentEntity, scene }, MI_ENTITY_STATIC);
			mi_collider_desc d = colliderDesc(c);
			if (!check(ctx, mi_collider_add(ctx.world, parent, &d, nullptr), "mi_collider_add")) { return false; }
		}
		if (!mirrorConstraints(ctx, scene, MI_CONSTRAINT_DISTANCE, ctx.distance) || !mirrorConstraints(ctx, scene, MI_CONSTRAINT_BALL, ctx.ball) || !mirrorConstraints(ctx, scene, MI_CONSTRAINT_FIXED, ctx.fixed) ||
			!mirrorConstraints(ctx, scene, MI_CONSTRAINT_HINGE, ctx.hinge) || !mirrorConstraints(ctx, scene, MI_CONSTRAINT_CONE_TWIST, ctx.coneTwist) || !mirrorConstraints(ctx, scene, MI_CONSTRAINT_SLIDER, ctx.slider)) { return false; }
		ctx.written.resize(nb);
		for (uint32 i = 0; i < nb; ++i)
		{
			scene_entity e = scene.getEntityFromComponentAtIndex<rigid_body_component>(i);
			const rigid_body_component& rb = scene.getComponentAtIndex<rigid_body_component>(i);
			const trs& t = physicsPose(e);
			ctx.written[i] = { t.position, t.rotation, rb.linearVelocity, rb.angularVelocity };
		}
		currentCounts(scene, ctx.counts);
		ctx.dirty = false;
		return true;
	}

	// game code between two steps: forces added to the accumulators, velocities set, bodies teleported
	bool pushHostEdits(mi_backend_context& ctx, game_scene& scene)
	{
		const uint32 nb = (uint32)ctx.written.size();
		std::vector<uint32> ids, forceIds; std::vector<float> states, forces, torques;
		for (uint32 i = 0; i < nb; ++i)
		{
			rigid_body_component& rb = scene.getComponentAtIndex<rigid_body_component>(i);
			scene_entity e = scene.getEntityFromComponentAtIndex<rigid_body_component>(i);
			const trs& t = physicsPose(e);
			const body_cache now = { t.position, t.rotation, rb.linearVelocity, rb.angularVelocity };
			if (memcmp(&now, &ctx.written[i], sizeof(body_cache)) != 0)
			{
				ids.push_back(ctx.bodyEntityIds[i]);
				const float s[13] = { now.position.x, now.position.y, now.position.z, now.rotation.x, now.rotation.y, now.rotation.z, now.rotation.w,
					now.linearVelocity.x, now.linearVelocity.y, now.linearVelocity.z, now.angularVelocity.x, now.angularVelocity.y, now.angularVelocity.z };
				states.insert(states.end(), s, s + 13);
			}
			if (rb.forceAccumulator.x != 0.f || rb.forceAccumulator.y != 0.f || rb.forceAccumulator.z != 0.f || rb.torqueAccumulator.x != 0.f || rb.torqueAccumulator.y != 0.f || rb.torqueAccumulator.z != 0.f)
			{
				forceIds.push_back(ctx.bodyEntityIds[i]);
				forces.insert(forces.end(), rb.forceAccumulator.data, rb.forceAccumulator.data + 3); torques.insert(torques.end(), rb.torqueAccumulator.data, rb.torqueAccumulator.data + 3);
				rb.forceAccumulator = vec3(0.f); rb.torqueAccumulator = vec3(0.f);       // the step consumes them (src/physics/rigid_body.cpp:139-140)
			}
		}
		if (!ids.empty() && !check(ctx, mi_world_set_body_states(ctx.world, (uint32)ids.size(), ids.data(), states.data()), "mi_world_set_body_states")) { return false; }
		if (!forceIds.empty() && !check(ctx, mi_entities_apply_forces(ctx.world, (uint32)forceIds.size(), forceIds.data(), forces.data(), torques.data()), "mi_entities_apply_forces")) { return false; }
		return true;
	}
}

void miOnPhysicsComponentChanged(entt::registry* registry)
{
	if (mi_backend_context* ctx = tryGetContextVariable<mi_backend_context>(*registry)) { ctx->dirty = true; }
}

mi_world* miBackendWorld(game_scene& scene)
{
	mi_backend_context& ctx = scene.createOrGetContextVariable<mi_backend_context>();
	uint32 now[12]; currentCounts(scene, now);
	if (ctx.dirty || !ctx.world || memcmp(now, ctx.counts, sizeof(now)) != 0) { if (!syncScene(ctx, scene)) { return nullptr; } }
	return ctx.world;
}

const char* miBackendError(game_scene& scene) { return scene.createOrGetContextVariable<mi_backend_context>().error.c_str(); }

// physicsStep (src/physics/physics.cpp:1364-1413) on the GPU.  The fixed-step accumulator, the <= maxPhysicsIterationsPerFrame sub-steps and the
// physics_transform0 / physics_transform1 interpolation happen inside mi_world_step; `timer` mirrors the library's accumulator for callers that look at it.
bool physicsStepMI355X(game_scene& scene, memory_arena& arena, float& timer, const physics_settings& settings, float dt)
{
	(void)arena;         // the library owns its device memory; the caller's arena is not touched
	mi_world* world = miBackendWorld(scene);
	if (!world) { return false; }
	mi_backend_context& ctx = scene.createOrGetContextVariable<mi_backend_context>();
	if (!pushHostEdits(ctx, scene)) { return false; }
	if (!resendEditedConstraints(ctx, scene, MI_CONSTRAINT_DISTANCE, ctx.distance) || !resendEditedConstraints(ctx, scene, MI_CONSTRAINT_BALL, ctx.ball) || !resendEditedConstraints(ctx, scene, MI_CONSTRAINT_FIXED, ctx.fixed) ||
		!resendEditedConstraints(ctx, scene, MI_CONSTRAINT_HINGE, ctx.hinge) || !resendEditedConstraints(ctx, scene, MI_CONSTRAINT_CONE_TWIST, ctx.coneTwist) || !resendEditedConstraints(ctx, scene, MI_CONSTRAINT_SLIDER, ctx.slider)) { return false; }

	mi_step_settings s = { settings.fixedFrameRate ? 1u : 0u, settings.frameRate, settings.maxPhysicsIterationsPerFrame, settings.numRigidSolverIterations };
	if (!check(ctx, mi_world_step(world, &s, dt), "mi_world_step")) { return false; }
	if (settings.fixedFrameRate)       // the accumulator of physics.cpp:1370-1394, for callers that read `timer`
	{
		const float fixedDt = 1.f / (float)settings.frameRate;
		timer += dt;
		uint32 it = 0;
		while (timer >= fixedDt && it++ < settings.maxPhysicsIterationsPerFrame) { timer -= fixedDt; }
		if (timer >= fixedDt) { timer = fmodf(timer, fixedDt); }
	}

	// write back: transform_component = the interpolated pose, physics_transform1 = the physics pose, velocities (what game code and the renderer read)
	uint32 n = 0; mi_world_num_entities(world, &n);
	static thread_local std::vector<float> pos, rot, ppos, prot, lin, ang;
	pos.resize(3 * (size_t)n); rot.resize(4 * (size_t)n); ppos.resize(3 * (size_t)n); prot.resize(4 * (size_t)n); lin.resize(3 * (size_t)n); ang.resize(3 * (size_t)n);
	// the poses come as views of the library's pinned rows when the step left them on the device (mi_world_view_*: produced there in this layout, one copy
	// the step enqueued itself); the copying calls cover the rest (nothing stepped, topology edits pending)
	const float *vPos = nullptr, *vRot = nullptr, *vPPos = nullptr, *vPRot = nullptr; uint32 vn = 0;
	if (mi_world_view_transforms(world, &vPos, &vRot, &vn) != MI_OK || vn != n)
	{
		if (!check(ctx, mi_world_get_transforms(world, pos.data(), rot.data(), n), "mi_world_get_transforms")) { return false; }
		vPos = pos.data(); vRot = rot.data();
	}
	if (mi_world_view_physics_transforms(world, &vPPos, &vPRot, &vn) != MI_OK || vn != n)
	{
		if (!check(ctx, mi_world_get_physics_transforms(world, ppos.data(), prot.data(), n), "mi_world_get_physics_transforms")) { return false; }
		vPPos = ppos.data(); vPRot = prot.data();
	}
	const float *vLin = nullptr, *vAng = nullptr;
	if (mi_world_view_velocities(world, &vLin, &vAng, &vn) != MI_OK || vn != n)
	{
		if (!check(ctx, mi_world_get_velocities(world, lin.data(), ang.data(), n), "mi_world_get_velocities")) { return false; }
		vLin = lin.data(); vAng = ang.data();
	}
	for (uint32 i = 0; i < (uint32)ctx.bodyEntityIds.size(); ++i)
	{
		const uint32 id = ctx.bodyEntityIds[i];
		scene_entity e = { ctx.entityOfIndex[id], scene };
		rigid_body_component& rb = scene.getComponentAtIndex<rigid_body_component>(i);
		memcpy(&rb.linearVelocity, &vLin[3 * (size_t)id], 12); memcpy(&rb.angularVelocity, &vAng[3 * (size_t)id], 12);
		if (transform_component* t = e.getComponentIfExists<transform_component>()) { memcpy(&t->position, &vPos[3 * (size_t)id], 12); memcpy(&t->rotation, &vRot[4 * (size_t)id], 16); }
		if (physics_transform1_component* p1 = e.getComponentIfExists<physics_transform1_component>()) { memcpy(&p1->position, &vPPos[3 * (size_t)id], 12); memcpy(&p1->rotation, &vPRot[4 * (size_t)id], 16); }
		const trs& t1 = physicsPose(e);
		ctx.written[i] = { t1.position, t1.rotation, rb.linearVelocity, rb.angularVelocity };
	}
	return true;
}
