// drive_reference_scenes.cpp — TEST INFRASTRUCTURE.  A program made of the REFERENCE's own scene and physics code (compiled from /root/reference by
// build_binding.py) plus the backend stub (physics_mi355x.cpp) plus libmi_physics.so:
//   * builds the reference's demo scene (src/application.cpp:183-251: a column of spheres over a trigger and the platform) and its learning scene
//     (src/learning/learned_locomotion.cpp:442-446: ground + humanoid_ragdoll::create, src/physics/ragdoll.cpp:12-123) TWICE with the reference's own
//     API — scene.createEntity(...).addComponent<transform_component / collider_component / rigid_body_component>(...), add*ConstraintFromGlobalPoints —
//   * steps one copy with the reference's physicsStep, the other with physicsStepMI355X (scene_entity::addComponent hooks -> stub -> mi_* C ABI),
//   * and compares what game code reads afterwards: transform_component, physics_transform1, velocities.
// Pass 1 (the library's own constraint order): bit-identical while nothing touches, the contact counts of every step equal while the two runs have
// not separated, poses close at the end.  Pass 2 (the reference's constraint order handed to the library every step: mi_debug_set_solve_order, from
// the contact list the instrumented physicsStepInternal hands out): bit-identical transforms and velocities, every step.
// Also exercised through the stub: a motor edited through the reference's mutable getConstraint() reference, a force added to rb.forceAccumulator,
// a velocity set by game code, an entity created mid-run (the mirror is rebuilt).  Exit code 0 = all of it held.
#include "physics/physics.h"
#include "physics/collision_broad.h"
#include "physics/ragdoll.h"
#include "scene/scene.h"
#include "mi_physics.h"
#include "mi_constraints.h"
#include "physics_mi355x.h"
#include <cstdio>

uint32 ref_sap_sorting_axis(game_scene& scene);      // instrumentation appended to the scratch copy of collision_broad.cpp (build_ref.py)

// ---- taps of the instrumented physicsStepInternal (build_ref.py): the reference's contact list of the step, in emission (= solve) order
static std::vector<uint32> g_pairs; static uint32 g_numContacts = 0, g_numCollisions = 0; static bool g_tap = false;
void ref_tap_broadphase(const collider_pair*, uint32) {}
void ref_tap_step(uint32, uint32, const bounding_box*, const collider_union*, uint32, uint32 numCollisions, uint32 numContacts, const collision_contact*, const constraint_body_pair*,
	const collider_pair* collidingPairs, const uint8*)
{
	if (!g_tap) { return; }
	g_pairs.clear();
	for (uint32 m = 0; m < numCollisions; ++m) { g_pairs.push_back(collidingPairs[m].colliderA); g_pairs.push_back(collidingPairs[m].colliderB); }
	g_numContacts = numContacts; g_numCollisions = numCollisions;
}

static const physics_material wood = { physics_material_type_wood, 0.1f, 0.5f, 1.f };

static void buildDemoScene(game_scene& scene, uint32 numSpheres)
{
	// src/application.cpp:196-217: spheres dropped in a column (mesh components are the renderer's; PHYSICS_ONLY builds have none)
	for (uint32 i = 0; i < numSpheres; ++i)
	{
		scene.createEntity("Sphere")
			.addComponent<transform_component>(vec3(25.f, 10.f + i * 3.f, -5.f), quat(vec3(0.f, 0.f, 1.f), deg2rad(1.f)), vec3(1.f))
			.addComponent<collider_component>(collider_component::asSphere({ vec3(0.f, 0.f, 0.f), 1.f }, wood))
			.addComponent<rigid_body_component>(false, 1.f);
	}
	// :226-228 the trigger volume
	scene.createEntity("Trigger")
		.addComponent<collider_component>(collider_component::asAABB(bounding_box::fromCenterRadius(vec3(25.f, 1.f, -5.f), vec3(5.f, 1.f, 5.f)), { physics_material_type_none, 0, 0, 0 }))
		.addComponent<trigger_component>(trigger_component{ [](trigger_event) {} });
	// :248-251 the platform
	scene.createEntity("Platform")
		.addComponent<transform_component>(vec3(10, -4.f, 0.f), quat(vec3(1.f, 0.f, 0.f), deg2rad(0.f)))
		.addComponent<collider_component>(collider_component::asAABB(bounding_box::fromCenterRadius(vec3(0.f, 0.f, 0.f), vec3(30.f, 4.f, 30.f)), { physics_material_type_metal, 0.1f, 1.f, 4.f }));
	// a few boxes and capsules beside the column, so that box and capsule routines and friction are on the path as well (same API)
	for (uint32 i = 0; i < 6; ++i)
	{
		scene.createEntity("Cube")
			.addComponent<transform_component>(vec3(18.f + 0.3f * i, 1.5f + i * 2.1f, 2.f), quat(vec3(0.f, 0.f, 1.f), deg2rad(3.f * i)))
			.addComponent<collider_component>(collider_component::asAABB(bounding_box::fromCenterRadius(vec3(0.f, 0.f, 0.f), vec3(1.f, 1.f, 2.f)), wood))
			.addComponent<rigid_body_component>(false, 1.f);
		scene.createEntity("Capsule")
			.addComponent<transform_component>(vec3(12.f, 2.f + i * 1.5f, -8.f + 0.2f * i), quat(vec3(0.f, 0.f, 1.f), deg2rad(80.f + i)))
			.addComponent<collider_component>(collider_component::asCapsule({ vec3(0.f, -0.6f, 0.f), vec3(0.f, 0.6f, 0.f), 0.3f }, wood))
			.addComponent<rigid_body_component>(false, 1.f);
	}
}

static humanoid_ragdoll buildRagdollScene(game_scene& scene)
{
	// src/learning/learned_locomotion.cpp:438-446
	physics_material groundMaterial = { physics_material_type_metal, 0.1f, 1.f, 4.f };
	scene.createEntity("Test ground")
		.addComponent<transform_component>(vec3(0.f, -4.f, 0.f), quat(vec3(1.f, 0.f, 0.f), deg2rad(0.f)))
		.addComponent<collider_component>(collider_component::asAABB(bounding_box::fromCenterRadius(vec3(0.f, 0.f, 0.f), vec3(20.f, 4.f, 20.f)), groundMaterial));
	return humanoid_ragdoll::create(scene, vec3(0.f, 1.25f, 0.f));
}

struct snapshot { std::vector<float> v; };
static snapshot take(game_scene& scene)
{
	snapshot s;
	const uint32 nb = scene.numberOfComponentsOfType<rigid_body_component>();
	for (uint32 i = 0; i < nb; ++i)
	{
		scene_entity e = scene.getEntityFromComponentAtIndex<rigid_body_component>(i);
		const rigid_body_component& rb = scene.getComponentAtIndex<rigid_body_component>(i);
		const transform_component& t = e.getComponent<transform_component>();
		const physics_transform1_component& p1 = e.getComponent<physics_transform1_component>();
		const float f[20] = { t.position.x, t.position.y, t.position.z, t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w, p1.position.x, p1.position.y, p1.position.z,
			p1.rotation.x, p1.rotation.y, p1.rotation.z, p1.rotation.w, rb.linearVelocity.x, rb.linearVelocity.y, rb.linearVelocity.z, rb.angularVelocity.x, rb.angularVelocity.y, rb.angularVelocity.z };
		s.v.insert(s.v.end(), f, f + 20);
	}
	return s;
}
static bool same(const snapshot& a, const snapshot& b) { return a.v.size() == b.v.size() && memcmp(a.v.data(), b.v.data(), a.v.size() * sizeof(float)) == 0; }
static float maxDiff(const snapshot& a, const snapshot& b) { float m = 0.f; for (size_t i = 0; i < a.v.size() && i < b.v.size(); ++i) { m = max(m, fabsf(a.v[i] - b.v[i])); } return a.v.size() == b.v.size() ? m : 1e30f; }
static float maxPositionDiff(const snapshot& a, const snapshot& b) { float m = 0.f; for (size_t i = 0; i + 20 <= a.v.size() && i + 20 <= b.v.size(); i += 20) { for (int k = 7; k < 10; ++k) { m = max(m, fabsf(a.v[i + k] - b.v[i + k])); } } return a.v.size() == b.v.size() ? m : 1e30f; }
static float lowestBody(const snapshot& a) { float m = 1e30f; for (size_t i = 0; i + 20 <= a.v.size(); i += 20) { m = min(m, a.v[i + 8]); } return m; }
static bool finite(const snapshot& a) { for (float f : a.v) { if (!(f == f) || fabsf(f) > 1e6f) { return false; } } return true; }

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAILED: " __VA_ARGS__); std::printf("\n"); } } while (0)

template <typename build_t, typename edit_t>
static void runScene(const char* name, build_t build, edit_t edit, uint32 steps, bool replay, bool hasJoints, float floorY)
{
	game_scene a, b;                    // a: the reference steps it; b: the library steps it through the stub
	memory_arena arenaA, arenaB; arenaA.initialize(0, GB(2)); arenaB.initialize(0, GB(1));
	auto handleA = build(a); auto handleB = build(b);
	physics_settings settings;          // the defaults of src/physics/physics.h:382-396 ...
	settings.simdBroadPhase = settings.simdNarrowPhase = settings.simdConstraintSolver = false;     // ... with the reference's SCALAR path (the AVX2 path is numerically looser by design)
	float timerA = 0.f, timerB = 0.f;
	const float dt = 1.f / (float)settings.frameRate;        // one internal step per call (the accumulator runs on both sides)
	uint32 firstContactStep = ~0u, identicalSteps = 0, countsEqualSteps = 0; bool separated = false; float lastDiff = 0.f, lastPosDiff = 0.f, lowest = 0.f;
	for (uint32 i = 0; i < steps; ++i)
	{
		edit(i, a, handleA); edit(i, b, handleB);
		const uint32 axis = ref_sap_sorting_axis(a);
		g_tap = true; physicsStep(a, arenaA, timerA, settings, dt); g_tap = false;
		if (g_numContacts && firstContactStep == ~0u) { firstContactStep = i; }
		mi_world* w = miBackendWorld(b);
		EXPECT(w != nullptr, "%s step %u: backend: %s", name, i, miBackendError(b));
		if (!w) { return; }
		if (replay) { mi_debug_set_sweep_axis(w, axis); mi_debug_set_solve_order(w, g_pairs.data(), (uint32)(g_pairs.size() / 2)); }
		const bool ok = physicsStepMI355X(b, arenaB, timerB, settings, dt);
		EXPECT(ok, "%s step %u: physicsStepMI355X: %s", name, i, miBackendError(b));
		if (!ok) { return; }
		mi_step_counts c; mi_world_get_counts(w, &c);
		const snapshot sa = take(a), sb = take(b);
		const bool identical = same(sa, sb);
		lastDiff = maxDiff(sa, sb); lastPosDiff = maxPositionDiff(sa, sb); lowest = lowestBody(sb);
		EXPECT(finite(sb), "%s step %u: a pose or velocity left the finite range", name, i);
		if (replay)
		{
			EXPECT(c.num_contacts == g_numContacts && c.num_collisions == g_numCollisions, "%s step %u (replay): contacts %u / %u, collisions %u / %u", name, i, c.num_contacts, g_numContacts, c.num_collisions, g_numCollisions);
			EXPECT(identical, "%s step %u (replay): transforms / velocities differ from the reference's (max %g)", name, i, lastDiff);
			if (!identical) { return; }
		}
		else
		{
			if (identical && !separated) { ++identicalSteps; } else { separated = true; }
			if (!separated || c.num_contacts == g_numContacts) { countsEqualSteps += c.num_contacts == g_numContacts; }
			// until something is solved in another order the two are the same computation (with joints that is never: they are solved from the first step)
			if (!hasJoints && i < firstContactStep) { EXPECT(identical, "%s step %u: not identical although nothing has been solved in another order yet", name, i); }
			if (!separated) { EXPECT(c.num_contacts == g_numContacts, "%s step %u: contacts %u / %u although the runs have not separated", name, i, c.num_contacts, g_numContacts); }
		}
		EXPECT(timerA == timerB, "%s step %u: the accumulators differ", name, i);
	}
	std::printf("%-22s %-14s steps %u, first contact at step %u, bit-identical steps %u, steps with equal contact counts %u; at the end: max position difference %g m, lowest body at y = %g\n",
		name, replay ? "reference order" : "own order", steps, firstContactStep, replay ? steps : identicalSteps, replay ? steps : countsEqualSteps, lastPosDiff, lowest);
	EXPECT(firstContactStep != ~0u, "%s: the scene never made contact", name);
	if (!replay)
	{
		// the library's own (colour-major) constraint order: the same physics along another PGS path — a pile / a falling ragdoll decorrelates,
		// so what is asserted is that it IS the same physics: identical while the order cannot matter, everything comes to rest on the ground, nothing tunnels
		if (!hasJoints) { EXPECT(identicalSteps >= firstContactStep, "%s: the runs separated before the first contact", name); }
		EXPECT(lowest > floorY - 0.05f, "%s: a body ended below the ground (y = %g)", name, lowest);
		EXPECT(lastPosDiff < 4.f, "%s: the two runs ended %g m apart", name, lastPosDiff);
	}
}

int main()
{
	for (int replay = 0; replay < 2; ++replay)
	{
		runScene("demo scene", [](game_scene& s) { buildDemoScene(s, 12); return 0; },
			[](uint32 i, game_scene& s, int)
			{
				if (i == 40)        // game code between two steps: a push through the accumulator, a velocity set directly (src/physics/physics.cpp:555-629 does the former)
				{
					rigid_body_component& rb = s.getComponentAtIndex<rigid_body_component>(3);
					rb.forceAccumulator += vec3(300.f, 0.f, 50.f); rb.torqueAccumulator += vec3(0.f, 40.f, 0.f);
					s.getComponentAtIndex<rigid_body_component>(5).linearVelocity = vec3(0.f, 2.f, -1.f);
				}
				if (i == 90)        // an entity created mid-run: the hook marks the mirror dirty, the stub rebuilds it
				{
					s.createEntity("Late sphere")
						.addComponent<transform_component>(vec3(24.f, 14.f, -4.f), quat::identity)
						.addComponent<collider_component>(collider_component::asSphere({ vec3(0.f, 0.f, 0.f), 0.7f }, wood))
						.addComponent<rigid_body_component>(false, 1.f);
				}
			}, 260, replay != 0, false, 0.f);
		runScene("ragdoll on the ground", [](game_scene& s) { return buildRagdollScene(s); },
			[](uint32 i, game_scene& s, humanoid_ragdoll& r)
			{
				if (i == 30)        // the ragdoll controller drives motors through the mutable reference getConstraint() returns (src/learning/learned_locomotion.cpp:222-260)
				{
					hinge_constraint& knee = getConstraint(s, r.leftKneeConstraint);
					knee.motorType = constraint_velocity_motor; knee.motorVelocity = 2.f; knee.maxMotorTorque = 200.f;
					cone_twist_constraint& shoulder = getConstraint(s, r.leftShoulderConstraint);
					shoulder.swingMotorType = constraint_velocity_motor; shoulder.swingMotorVelocity = 1.f; shoulder.maxSwingMotorTorque = 100.f; shoulder.swingMotorAxis = 0.5f;
				}
			}, 240, replay != 0, true, 0.f);
	}
	std::printf(failures ? "BINDING CHECK FAILED (%d)\n" : "BINDING CHECK OK\n", failures);
	return failures ? 1 : 0;
}
