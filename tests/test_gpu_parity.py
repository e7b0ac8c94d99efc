"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle replaying the
same canonical schedule.  Integers (pairs, manifolds, contacts, colours) must be bit-exact; poses and
velocities are compared bit-exactly too (stricter than the 1e-4 relative tolerance north_star allows:
with FMA contraction off and IEEE div/sqrt the two paths perform identical fp32 operations)."""
import sys
from pathlib import Path
import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import make_golden  # noqa: E402
from d3d12renderer_amd import scenes, capi  # noqa: E402
from helpers import contact_set, single_body_scene  # noqa: E402

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4   # north_star tolerance for float state; the assertions below are stricter (bit-exact)


def gpu_world(mi_lib):
    return mi_lib.create_world(0)


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_gpu_matches_golden_bit_exact(mi_lib, name):
    make, steps = make_golden.CASES[name]
    sc = make()
    w = sc.populate(gpu_world(mi_lib))
    s = sc.settings()
    counts = []
    for _ in range(steps):
        w.step_fixed(s, sc.dt, 1)
        c = w.counts()
        counts.append([c["num_broadphase_overlaps"], c["num_collisions"], c["num_contacts"], c["num_colors"], c["sorting_axis"]])
    want = np.load(ROOT / "tests" / "golden" / f"{name}_canonical.npz")
    assert np.array_equal(np.asarray(counts, np.uint32), want["counts"])
    p, q = w.physics_transforms(); v, a = w.velocities()
    for got, k in ((p, "pos"), (q, "rot"), (v, "lin"), (a, "ang")):
        assert np.allclose(got, want[k], rtol=REL_TOL, atol=1e-6), k
        assert got.tobytes() == want[k].tobytes(), k


@pytest.mark.parametrize("make,steps", [
    (lambda: scenes.sphere_drop(10), 130),
    (lambda: scenes.mixed_stack(12, 6, 12), 80),
    (lambda: scenes.obb_pile(12, 8, 12, spacing=1.1), 80),
    (lambda: scenes.shape_zoo(8, 5, 8), 150),
    (lambda: scenes.ragdolls(4, 4), 160),
    (lambda: scenes.joint_zoo(copies=3), 200),
    (lambda: scenes.vehicles(3, 2), 160),
    (lambda: scenes.terrain_field(10, 2, 10), 260),     # heightmap terrain: quadtree walk + triangle tests + lowest-point contacts
    (lambda: scenes.terrain_wide_colliders(), 300),     # colliders spanning hundreds of terrain cells and chunk borders (k_hm_contacts' large-window instance) among small ones
])
@pytest.mark.parametrize("stepping", ["speculative", "synchronous"])
def test_gpu_vs_oracle_trajectory_and_contacts(mi_lib, oracle_mod, monkeypatch, make, steps, stepping):
    """Every scene type against the oracle, step by step; `synchronous` (MI_ASYNC=0): every step sized from read-backs inside the step — the path the first step of a
    world and every re-run of a void speculative step take (joints through the per-type launches, triggers on the host, terrain counts read back, exact tile tables)."""
    if stepping == "synchronous":
        monkeypatch.setenv("MI_ASYNC", "0")
    sc = make()
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    for i in range(steps):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        assert g.counts() == o.counts(), f"step {i}"
        if i % 20 == 0 or i == steps - 1:
            assert contact_set(g.contacts()) == contact_set(o.contacts()), f"step {i}"
    pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
    assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes()
    assert g.counts()["num_contacts"] > 0


@pytest.mark.parametrize("make,steps", [
    (lambda: scenes.shape_zoo(8, 5, 8), 150),             # every collider type, GJK / EPA manifolds, mixed contact counts
    (lambda: scenes.terrain_field(10, 2, 10), 260),       # terrain contacts: one-contact manifolds against a virtual static body
    (lambda: scenes.mixed_stack(12, 6, 12), 80),
])
def test_gpu_xcd_partitioned_solver_on_other_scene_types(mi_lib, oracle_mod, monkeypatch, make, steps):
    """The XCD-partitioned persistent solver (forced on for these small scenes; default from 16 384 manifolds) with the other
    manifold sources: same trajectory as the oracle, bit for bit."""
    monkeypatch.setenv("MI_PERSIST_XCD_MIN", "1")
    sc = make()
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    kinds = set()
    for i in range(steps):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        assert g.counts() == o.counts(), f"step {i}"
        kinds.add(g.solver_kind())
    pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
    assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes()
    assert 4 in kinds, kinds


def test_gpu_contact_set_equals_reference_order_oracle_first_step(mi_lib, oracle_mod):
    """The GPU's grid broad phase + canonical orientation must reproduce the SAP pipeline's manifolds."""
    sc = scenes.obb_pile(10, 4, 10, spacing=1.0)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_REFERENCE))
    g.step_fixed(sc.settings(), sc.dt, 1); o.step_fixed(sc.settings(), sc.dt, 1)
    cg, co = g.counts(), o.counts()
    for k in ("num_broadphase_overlaps", "num_collisions", "num_contacts"):
        assert cg[k] == co[k] and cg[k] > 0
    assert contact_set(g.contacts()) == contact_set(o.contacts())
    assert np.array_equal(g.aabbs(), o.aabbs())


def test_gpu_physics_step_accumulator(mi_lib, oracle_mod):
    sc = single_body_scene(capi.SPHERE, (0, 0, 0, 1.0), pos=(0, 5, 0))
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = capi.StepSettings(1, 120, 4, 10)
    for dt in (1 / 240, 1 / 240 + 1 / 480, 1 / 60, 0.1, 1 / 120):
        g.step(s, dt); o.step(s, dt)
        pg, qg = g.transforms(); po, qo = o.transforms()
        assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes()


@pytest.mark.parametrize("graphs", ["default", "force"])
@pytest.mark.parametrize("make", [lambda: scenes.obb_pile(12, 5, 12, spacing=1.05), lambda: scenes.ragdolls(3, 3), lambda: scenes.vehicles(2, 2)], ids=["pile", "ragdolls (joint islands)", "vehicles on hull tiles"])
def test_gpu_pose_rows_for_a_caller_that_reads_them_after_every_step(mi_lib, oracle_mod, monkeypatch, graphs, make):
    """The renderer's pattern: physicsStep, then the transform of every entity, frame after frame.  The rows are produced on the device in the
    caller's layout (k_entity_poses: lerp / nlerp of physics_transform0 and 1, or physics_transform1 itself) and come over in one copy which the
    step enqueues itself once somebody has asked after the previous step.  Same bytes as the oracle's transforms, as the copying call and as the
    per-array path (MI_POSE_STREAM=0); entities without a rigid body (the static ground) keep the host's transform; a view stays intact for one
    more step."""
    if graphs == "force":
        monkeypatch.setenv("MI_GRAPH", "force")      # the steps of this small scene replayed as HIP graphs: the rows are enqueued behind the graph launch, outside any capture
    sc = make()
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    monkeypatch.setenv("MI_POSE_STREAM", "0")
    h = sc.populate(gpu_world(mi_lib))
    monkeypatch.delenv("MI_POSE_STREAM")
    s = sc.settings()
    assert g.num_entities() > sc.num_bodies, "the scene has entities without a rigid body"
    kept = None
    for i in range(24):
        dt = sc.dt * (1.15, 0.55, 0.75, 0.95, 1.35)[i % 5]   # 0, 1 or 2 internal steps per call, a different interpolation factor every time
        for w in (g, o, h):
            w.step(s, dt)
        vp, vr = g.transforms_view()
        if kept is not None:                          # the view handed out one step ago is still what it was
            assert kept[0].tobytes() == kept[2] and kept[1].tobytes() == kept[3]
        kept = (vp, vr, vp.tobytes(), vr.tobytes())
        p, r = g.transforms(); po, ro = o.transforms(); ph, rh = h.transforms()
        assert p.tobytes() == po.tobytes() == ph.tobytes() == vp.tobytes(), f"call {i}"
        assert r.tobytes() == ro.tobytes() == rh.tobytes() == vr.tobytes(), f"call {i}"
        if i % 4 == 3:
            pp, pr = g.physics_transforms(); qp, qr = o.physics_transforms(); wp, wr = g.transforms_view(physics=True)
            assert pp.tobytes() == qp.tobytes() == wp.tobytes() and pr.tobytes() == qr.tobytes() == wr.tobytes()
        if i >= 8:                                    # ... and from here on the caller reads the velocities every frame too: they ride in the same rows
            lv, av = g.velocities(); lo, ao = o.velocities(); lw, aw = g.velocities_view()
            assert lv.tobytes() == lo.tobytes() == lw.tobytes() and av.tobytes() == ao.tobytes() == aw.tobytes(), f"call {i}"
            assert vp.tobytes() == p.tobytes()       # (the view of the poses is still what it was)
    ahead, on_demand = g.pose_stream_stats()
    assert ahead >= 12 and h.pose_stream_stats() == (0, 0)
    for _ in range(6):                                # plain internal steps: transform = physics_transform1
        for w in (g, o):
            w.step_fixed(s, sc.dt, 2)
        assert g.transforms()[1].tobytes() == o.transforms()[1].tobytes() and g.physics_transforms()[0].tobytes() == o.physics_transforms()[0].tobytes()
    assert g.pose_stream_stats()[0] > ahead
    ids = np.arange(sc.num_bodies, dtype=np.uint32)  # a state written from outside between two reads: the rows follow it
    st = o.get_body_states(ids); st[:, 0] += 0.25
    g.set_body_states(ids, st); o.set_body_states(ids, st)
    assert g.physics_transforms()[0].tobytes() == o.physics_transforms()[0].tobytes() and g.transforms_view(physics=True)[0].tobytes() == o.physics_transforms()[0].tobytes()
    g.step(s, 0.0); o.step(s, 0.0)                    # physicsStep settles what the plain steps left pending (a full download) and runs no internal step: nothing newer on the device, no view to hand out
    with pytest.raises(capi.PhysicsError):
        g.transforms_view()
    assert g.transforms()[0].tobytes() == o.transforms()[0].tobytes()


def test_gpu_pose_rows_one_frame_behind(mi_lib, oracle_mod):
    """mi_world_view_transforms_landed: the newest pose rows that are COMPLETE in host memory, without waiting for a copy still on the bus.  Whatever step the call says the
    rows belong to, they are that step's physics transforms bit for bit (the oracle's, recorded step by step); the step is the last one or the one before; a renderer's loop
    (step, view) never falls further behind and the rows it was handed stay intact until its next stepping call."""
    sc = scenes.obb_pile(16, 6, 16, spacing=1.05)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    want = {}
    behind = []
    for i in range(40):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        po, ro = o.physics_transforms(); want[i + 1] = (po.tobytes(), ro.tobytes())
        p, r, of = g.transforms_view_landed(physics=True)
        assert of in (i, i + 1) and of >= 1, (i, of)
        assert p.tobytes() == want[of][0] and r.tobytes() == want[of][1], f"call {i}: the rows of step {of}"
        snap = (p.tobytes(), r.tobytes())
        _ = g.counts()                                # (host work between the view and the next step)
        assert (p.tobytes(), r.tobytes()) == snap     # intact until the next stepping call
        behind.append(i + 1 - of)
    assert g.pose_stream_stats()[0] >= 30             # the steps enqueue the rows themselves
    # the blocking view still hands out the current step's rows
    vp, vr = g.transforms_view(physics=True)
    assert vp.tobytes() == want[40][0] and vr.tobytes() == want[40][1]


def test_gpu_pose_and_velocity_readbacks_agree_with_a_full_download(mi_lib, oracle_mod):
    """mi_world_get_transforms / _physics_transforms / _velocities read straight from the device (2-4 arrays, host mirror untouched); everything
    else goes through a full download of the body state.  Both give the same bytes, in any order, interpolated or not, and equal the oracle."""
    sc = scenes.mixed_stack(6, 4, 6)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings(); ids = np.arange(sc.num_bodies, dtype=np.uint32)
    for i, dt in enumerate((1 / 120, 1 / 240 + 1 / 480, 1 / 60, 1 / 480, 0.03)):
        g.step(s, dt); o.step(s, dt)
        fast = (g.transforms(), g.physics_transforms(), g.velocities())          # device -> caller, the mirror stays stale
        again = (g.transforms(), g.physics_transforms(), g.velocities())         # idempotent
        st = g.get_body_states(ids)                                             # forces the full download (and settles a pending interpolation)
        slow = (g.transforms(), g.physics_transforms(), g.velocities())          # now from the host mirror
        ref = (o.transforms(), o.physics_transforms(), o.velocities())
        for a, b, c, d in zip(fast, again, slow, ref):
            for k in range(2):
                assert a[k].tobytes() == b[k].tobytes() == c[k].tobytes() == d[k].tobytes(), f"call {i}"
        assert st.tobytes() == o.get_body_states(ids).tobytes()
    for _ in range(3):                                                           # and after plain internal steps (transform = physics_transform1)
        g.step_fixed(s, sc.dt, 2); o.step_fixed(s, sc.dt, 2)
        assert g.transforms()[0].tobytes() == o.transforms()[0].tobytes() and g.velocities()[1].tobytes() == o.velocities()[1].tobytes()
        g.get_body_states(ids)
        assert g.transforms()[1].tobytes() == o.transforms()[1].tobytes()


@pytest.mark.parametrize("name", sorted(scenes.EDGE_CASES))
def test_gpu_degenerate_scenes_match_oracle(mi_lib, oracle_mod, name):
    """scenes.EDGE_CASES (exactly aligned faces, parallel capsule / cylinder branches, kinematic platform, compound bodies, odd body
    parameters, a world without contacts): the same scenes tests/test_reference_pin.py steps through the reference itself."""
    sc = scenes.EDGE_CASES[name]()
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    for i in range(180):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        assert g.counts() == o.counts(), f"step {i}"
        assert contact_set(g.contacts()) == contact_set(o.contacts()), f"step {i}"
        pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
        assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes(), f"step {i}"


def test_gpu_edge_cases(mi_lib, oracle_mod):
    # empty world steps; bodies without colliders integrate and take external forces
    w = gpu_world(mi_lib)
    w.step_fixed(capi.StepSettings(), 1 / 120, 1)
    assert w.counts()["num_rigid_bodies"] == 0
    e = scenes.make_entities(2)
    e["position"][1] = (3, 0, 0)
    out = []
    for w in (gpu_world(mi_lib), oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)):
        w.create_entities(e)
        w.apply_force(0, force=(10, 0, 0), torque=(0, 1, 0))
        w.step_fixed(capi.StepSettings(), 1 / 120, 3)
        out.append((w.physics_transforms()[0].tobytes(), w.physics_transforms()[1].tobytes(), w.velocities()[0].tobytes()))
    assert out[0] == out[1]
    # adding bodies between steps (topology change -> re-upload) keeps parity
    sc = scenes.sphere_drop(4)
    ws = [sc.populate(gpu_world(mi_lib)), sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))]
    extra = scenes.make_entities(1); extra["position"][0] = (0.3, 30, 0.2)
    col = scenes.make_colliders(1, capi.SPHERE); col["shape"][0, 3] = 0.7
    res = []
    for w in ws:
        w.step_fixed(sc.settings(), sc.dt, 60)
        first = w.create_entities(extra)
        w.add_colliders([first], col)
        w.step_fixed(sc.settings(), sc.dt, 120)
        res.append((w.physics_transforms()[0].tobytes(), w.counts()))
    assert res[0] == res[1]


def test_gpu_collision_events_match_oracle(mi_lib, oracle_mod):
    """collisionBegin / collisionEnd events (polled): same events, same order, bit-identical payloads, every step."""
    sc = scenes.mixed_stack(6, 5, 6)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    g.enable_events(); o.enable_events()
    s = sc.settings()
    total = ends = 0
    for i in range(150):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        eg, eo = g.poll_events(), o.poll_events()
        assert eg.tobytes() == eo.tobytes(), f"step {i}: {len(eg)} vs {len(eo)} events"
        total += len(eg); ends += int((eg["type"] == capi.EVENT_COLLISION_END).sum())
    assert total > 100 and ends > 0
    pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
    assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes()


def test_gpu_triggers_and_force_fields_match_oracle(mi_lib, oracle_mod):
    """handleNonCollisionInteractions (physics.cpp:952-1039): localized + global force fields and trigger enter / leave events
    over every collider type, bit-exact against the oracle's canonical schedule, every step."""
    sc = scenes.zones()
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    g.enable_events(); o.enable_events()
    s = sc.settings()
    enters = leaves = 0
    for i in range(240):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        eg, eo = g.poll_events(), o.poll_events()
        assert eg.tobytes() == eo.tobytes(), f"step {i}: {len(eg)} vs {len(eo)} events"
        enters += int((eg["type"] == capi.EVENT_TRIGGER_ENTER).sum()); leaves += int((eg["type"] == capi.EVENT_TRIGGER_LEAVE).sum())
        if i % 40 == 0:
            vg, wg = g.velocities(); vo, wo = o.velocities()
            assert vg.tobytes() == vo.tobytes() and wg.tobytes() == wo.tobytes(), f"step {i}"
    assert enters > 10 and leaves > 5
    pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
    assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes()
    # without events the forces still apply and the state stays identical
    g2 = sc.populate(gpu_world(mi_lib)); o2 = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    g2.step_fixed(s, sc.dt, 60); o2.step_fixed(s, sc.dt, 60)
    assert g2.physics_transforms()[0].tobytes() == o2.physics_transforms()[0].tobytes()


def test_gpu_heightmap_full_size_properties(mi_lib):
    """65 536 mixed bodies on a 4 x 4-chunk heightmap (too slow for the oracle): deterministic across runs, finite state,
    nothing ends up under the terrain surface, and the steps after the first run speculatively."""
    sc = scenes.terrain_big()
    res = []
    for _ in range(2):
        w = sc.populate(gpu_world(mi_lib))
        w.step_fixed(sc.settings(), sc.dt, 120)
        p, q = w.physics_transforms()
        res.append((p.tobytes(), q.tobytes(), w.counts()))
    assert res[0] == res[1]
    assert np.isfinite(p).all() and np.isfinite(q).all()
    c = w.counts()
    assert c["num_contacts"] > c["num_collisions"] > 10000
    on_map = (np.abs(p[:, 0]) < 79.0) & (np.abs(p[:, 2]) < 79.0)
    idx = np.flatnonzero(on_map)[::97]
    h = np.array([w.heightmap_height(float(p[i, 0]), float(p[i, 2])) for i in idx])
    assert (p[idx, 1] > h - 0.3).all()
    total, spec, retries = w.step_mode_stats()
    assert spec >= total - 1 - retries and retries <= 20   # everything lands within a few steps: the contact count jumps past the speculative bounds


def test_gpu_ray_interactions_and_batched_edits_match_oracle(mi_lib, oracle_mod):
    """testPhysicsInteraction over every collider type (sphere, capsule, cylinder, AABB, OBB, hull), many rays per call, with
    and without entity ranges, interleaved with steps; batched force application and batched constraint (motor) updates on
    the fast path (no topology re-upload): GPU state stays bit-identical to the oracle's."""
    sc = scenes.shape_zoo(6, 3, 6)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    nb = sc.num_bodies
    rng = np.random.default_rng(5)
    for it in range(40):
        g.step_fixed(s, sc.dt, 3); o.step_fixed(s, sc.dt, 3)
        p, _ = o.physics_transforms()
        nrays = 24
        target = p[rng.integers(0, nb, nrays)] + rng.uniform(-0.2, 0.2, (nrays, 3)).astype(np.float32)
        direction = rng.normal(size=(nrays, 3)).astype(np.float32); direction /= np.linalg.norm(direction, axis=1, keepdims=True)
        origin = (target - 6.0 * direction).astype(np.float32)
        strength = rng.uniform(50, 400, nrays).astype(np.float32)
        lo = rng.integers(0, nb, nrays); ranges = np.stack([lo, lo + rng.integers(1, 40, nrays)], axis=1).astype(np.uint32)
        use_ranges = ranges if it % 2 else None
        g.test_interactions(origin, direction, strength, use_ranges); o.test_interactions(origin, direction, strength, use_ranges)
        ents = rng.integers(0, nb, 16).astype(np.uint32)
        f = rng.uniform(-20, 20, (16, 3)).astype(np.float32); t = rng.uniform(-2, 2, (16, 3)).astype(np.float32)
        g.apply_forces(ents, f, t); o.apply_forces(ents, f, t)
    steps, spec, retries = g.step_mode_stats()
    assert spec >= steps - 2 - retries                     # the edits did not throw the world off its fast path
    pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
    vg, wg = g.velocities(); vo, wo = o.velocities()
    assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes() and vg.tobytes() == vo.tobytes() and wg.tobytes() == wo.tobytes()
    # motors through the batched update: ragdoll hinges driven to a target angle
    sc = scenes.ragdolls(3, 3)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    nh = 6 * 9
    for it in range(30):
        pods = np.concatenate([g.get_constraint(capi.CONSTRAINT_HINGE, i) for i in range(nh)])
        pods["motor_type"] = 1; pods["max_motor_torque"] = 200.0; pods["motor_velocity_or_target_angle"] = 0.3 * np.sin(0.2 * it)
        g.update_constraints(capi.CONSTRAINT_HINGE, np.arange(nh), pods); o.update_constraints(capi.CONSTRAINT_HINGE, np.arange(nh), pods)
        g.step_fixed(s, sc.dt, 2); o.step_fixed(s, sc.dt, 2)
    assert g.physics_transforms()[0].tobytes() == o.physics_transforms()[0].tobytes()
    steps, spec, retries = g.step_mode_stats()
    assert spec >= steps - 2 - retries


def test_gpu_constraint_deletion_matches_oracle(mi_lib, oracle_mod):
    """deleteConstraint / deleteAllConstraintsFromEntity / deleteAllConstraints while the simulation runs: the pool order after
    EnTT's swap-and-pop decides the joint colouring priorities and the island programs, so GPU and oracle must reorder alike."""
    sc = scenes.ragdolls(4, 3)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    both = (g, o)
    def run(n):
        for i in range(n):
            for w in both: w.step_fixed(s, sc.dt, 1)
            assert g.counts() == o.counts()
        assert g.physics_transforms()[0].tobytes() == o.physics_transforms()[0].tobytes()
    run(30)
    for w in both:                                   # scattered single deletions: knees, elbows, a few cone twists
        for cid in (2, 9, 17, 40, 41, 65):
            w.destroy_constraint(capi.CONSTRAINT_HINGE, cid)
        for cid in (0, 8, 30, 83):
            w.destroy_constraint(capi.CONSTRAINT_CONE_TWIST, cid)
    run(40)
    for w in both:                                   # every constraint of three torsos (entities 0, 14, 28)
        for ent in (0, 14, 28):
            w.destroy_entity_constraints(ent)
    run(40)
    pod = g.get_constraint(capi.CONSTRAINT_HINGE, 71)
    for w in both:                                   # handles stay valid: re-create one and edit another
        assert w.add_constraint(capi.CONSTRAINT_HINGE, 14 * 11 + 2, 14 * 11 + 3, pod) == 72
        p = w.get_constraint(capi.CONSTRAINT_HINGE, 70); p["motor_type"] = 1; p["max_motor_torque"] = 90.0; p["motor_velocity_or_target_angle"] = 0.4
        w.update_constraint(capi.CONSTRAINT_HINGE, 70, p)
    run(30)
    for w in both:
        w.destroy_all_constraints()
    run(30)


@pytest.mark.parametrize("iters", [(0, 1, 0), (2, 3, 1)])
def test_gpu_cloth_matches_oracle(mi_lib, oracle_mod, iters):
    """cloth_component on the device (one workgroup per cloth, 12-colour Gauss-Seidel passes, per-vertex wind gather) against
    the oracle's canonical order, bit for bit: three cloths of different sizes next to a running rigid-body scene, wind from a
    global force field, fixed row moved (rigidly and not), properties edited mid-run."""
    sc = scenes.mixed_stack(4, 3, 4)
    worlds = []
    for make in (lambda: gpu_world(mi_lib), lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)):
        w = sc.populate(make())
        e = scenes.make_entities(1, capi.ENTITY_FORCE_FIELD)
        w.create_entities(e); w.set_force(sc.entities.shape[0], (1.5, 0.2, 4.0))     # global wind (no colliders)
        w.set_cloth_iterations(*iters)
        worlds.append(w)
    g, o = worlds
    shapes = [(9, 7, 2.0, 1.5, 3.0), (32, 32, 4.0, 4.0, 10.0), (40, 25, 5.0, 3.0, 6.0)]
    ids = []
    for w in worlds:
        ids = [w.create_cloth(wd, ht, gx, gy, m, stiffness=0.5 + 0.1 * k, damping=0.3 + 0.2 * k) for k, (gx, gy, wd, ht, m) in enumerate(shapes)]
        w.set_cloth_fixed_vertices(ids[0], (0.0, 4.0, 0.0), move_rigid=True)
        w.set_cloth_fixed_vertices(ids[1], (6.0, 5.0, 0.0), scenes.q_axis_angle((0, 1, 0), 0.7), move_rigid=True)
        w.set_cloth_fixed_vertices(ids[2], (-6.0, 5.0, 1.0), scenes.q_axis_angle((1, 0, 0), -0.4))
    s = sc.settings()
    def check(tag):
        for c, (gx, gy, *_r) in zip(ids, shapes):
            pg, vg = g.cloth_state(c, gx * gy); po, vo = o.cloth_state(c, gx * gy)
            assert np.isfinite(pg).all() and pg.tobytes() == po.tobytes() and vg.tobytes() == vo.tobytes(), f"{tag}: cloth {c}"
    check("initial")
    for it in range(6):
        for w in worlds: w.step_fixed(s, sc.dt, 25)
        check(f"block {it}")
        if it == 2:
            for w in worlds:
                w.set_cloth_properties(ids[1], 14.0, 0.8, 0.9, 0.7)        # recalculateProperties on the next step
                w.set_cloth_fixed_vertices(ids[0], (0.5, 4.2, 0.3))         # drag the locked row
    assert g.physics_transforms()[0].tobytes() == o.physics_transforms()[0].tobytes()
    # cloth alone (no rigid bodies at all) still steps
    g2 = gpu_world(mi_lib); o2 = oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)
    for w in (g2, o2):
        c = w.create_cloth(2.0, 2.0, 12, 12, 2.0)
        w.step_fixed(s, sc.dt, 40)
    assert g2.cloth_state(0, 144)[0].tobytes() == o2.cloth_state(0, 144)[0].tobytes()


@pytest.mark.parametrize("env,kind", [({}, 5), ({"MI_SOLVER": "persist"}, 5), ({"MI_SOLVER": "flow"}, 1), ({"MI_SOLVER": "persist-global"}, 5), ({"MI_PERSIST_XCD": "0"}, 2), ({"MI_PERSIST_XCD_SINGLE": "0"}, 2),
                                      ({"MI_PERSIST_XCD_SINGLE": "0", "MI_SOLVER": "persist-global"}, 2), ({"MI_PERSIST_XCD_SINGLE": "0", "MI_SOLVER": "persist-granules"}, 2),
                                      ({"MI_PERSIST_XCD_MIN": "1"}, 4), ({"MI_PERSIST_XCD_MIN": "1", "MI_SOLVER": "persist-global"}, 4),
                                      ({"MI_PERSIST_RESIDENT": "0"}, 5), ({"MI_PERSIST_XCD_MIN": "1", "MI_PERSIST_RESIDENT": "0"}, 4), ({"MI_PERSIST_XCD": "0", "MI_PERSIST_RESIDENT": "0"}, 2),
                                      ({"MI_SOLVER": "persist-granules"}, 5), ({"MI_PERSIST_XCD_MIN": "1", "MI_SOLVER": "persist-granules"}, 4),
                                      ({"MI_PERSIST_XCD_MIN": "1", "MI_PERSIST_XCD_FAULT": "1"}, 2), ({"MI_PERSIST_XCD_FAULT": "1"}, 2), ({"MI_READBACK": "copy"}, 5),
                                      ({"MI_PERSIST_WAVES": "8"}, 2), ({"MI_PERSIST_WAVES": "2"}, 2),
                                      ({"MI_SOLVER": "flow", "MI_FLOW_FAULT": "1"}, 0), ({"MI_ASYNC": "0"}, None)])
def test_gpu_other_contact_solvers_match_oracle(mi_lib, oracle_mod, monkeypatch, env, kind):
    """Every dataflow contact solver gives the same results, bit for bit (which lane / wave / XCD runs a slot is invisible to the
    body-version dataflow):  MI_SOLVER=flow -> k_contact_solve_flow (one workgroup per (sweep, tile), dispatch-ordered; also the
    automatic fallback and the path taken with joints);  persist-global -> the persistent kernel with the slot data read from
    global memory instead of LDS (what piles beyond ~500 k manifolds get);  persist-granules -> the accumulated impulses as
    tagged granules in memory as well (piles beyond ~1.2 M manifolds);  MI_PERSIST_XCD_MIN=1 -> XCD partitioning (spatially
    sorted slots, per-XCD tile lists, XCD-local bodies through L2) even on this small pile (default from 16384 manifolds up);
    MI_PERSIST_XCD=0 -> never partitioned;  no variable at all -> a pile this small runs on ONE XCD (kind 5: all tiles in XCD 0's list, every
    body hand-over through its L2); MI_PERSIST_XCD_SINGLE=0 -> small piles on all XCDs, bodies through memory;  MI_PERSIST_XCD_FAULT -> one workgroup reports that blockIdx % 8 did not identify its
    XCD: the step is re-run from untouched state and the world continues unpartitioned;  MI_READBACK=copy -> the end-of-step
    read-back as an async copy + stream synchronise instead of the kernel-published record the host spins on;
    MI_FLOW_FAULT -> the dispatch-ordered kernel reports an exhausted spin budget once (what a shared device can cause): the step
    is re-run from untouched state with one launch per colour (solver kind 0) and stays there for the next 256 steps;
    MI_PERSIST_WAVES=8 / 2 -> so few persistent workgroups that each owns many tiles: the library itself then moves first the slot
    data and then the impulses out of LDS (the choices it makes for piles of 0.5 M / 1.2 M manifolds and more);
    MI_PERSIST_RESIDENT=0 -> every tile's rows stream through the ring (default: the rows of a wave's first tiles stay resident in registers)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = scenes.obb_pile(14, 8, 14, spacing=1.05)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    for i in range(70):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        assert g.counts() == o.counts(), f"step {i}"
    assert g.counts()["num_contacts"] > 3000
    assert g.physics_transforms()[0].tobytes() == o.physics_transforms()[0].tobytes()
    assert kind is None or g.solver_kind() == kind   # (MI_ASYNC=0: every step synchronous, whatever solver the exact sizes select)
    if kind is None: assert g.step_mode_stats()[1] == 0
    assert g.step_mode_stats()[2] <= 2, "the partitioned solver must not keep falling back"
    # timing is opt-in: nothing by default; level 2 = the whole step and the solve stage; level 1 = every stage
    t = g.stage_times()
    assert t["total"] == 0 and t["solve"] == 0
    g.set_stage_timing(2)
    g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
    t = g.stage_times()
    assert t["total"] > 0 and t["solve"] > 0 and t["broadphase"] == 0
    g.set_stage_timing(1)
    g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
    t = g.stage_times()
    assert t["broadphase"] > 0 and t["narrowphase"] > 0 and abs(t["total"] - sum(v for k, v in t.items() if k != "total")) < 0.2 * t["total"]
    assert g.physics_transforms()[0].tobytes() == o.physics_transforms()[0].tobytes()


@pytest.mark.parametrize("env", [{"MI_FUSE_RESET": "0"}, {"MI_ROUND0_EMIT": "0"}, {"MI_FUSE_LARGE": "0"}, {"MI_FINISH_IN_NARROW": "0"}, {"MI_COLOR_TAIL": "0"},
                                 {"MI_COLOR_ROUNDS_MAX": "1"}, {"MI_COLOR_ROUNDS_MAX": "1", "MI_ROUND0_EMIT": "0"}, {"MI_COLOR_ROUNDS_MAX": "2", "MI_PERSIST_XCD_MIN": "1"},
                                 {"MI_FUSE_KEYS": "0", "MI_PERSIST_XCD_MIN": "1"},
                                 {"MI_FUSE_RESET": "0", "MI_ROUND0_EMIT": "0", "MI_FUSE_LARGE": "0", "MI_FINISH_IN_NARROW": "0", "MI_COLOR_TAIL": "0", "MI_FUSE_KEYS": "0", "MI_PERSIST_XCD_MIN": "1"}],
                         ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()))
def test_gpu_step_launch_variants_match_oracle(mi_lib, oracle_mod, monkeypatch, env):
    """The launches a steady step no longer makes, each behind a switch that brings it back (knobs.hpp): k_reset_scalars (its work rides in k_publish_readback), colouring
    round 0 (inside k_emit_manifolds), the large colliders' pair pass (first workgroups of k_bp_pairs), k_pair_finish (every workgroup of k_narrow derives the list's final
    counts itself), the margin of colouring rounds (k_bin_hist runs whatever rounds the enqueued ones left over: MI_COLOR_ROUNDS_MAX=1 makes it run nearly all of them),
    k_manifold_keys (same launch as k_integrate_forces: k_forces_keys; needs the XCD-partitioned layout, forced onto this small pile).  A falling pile — every count grows from step
    to step — against the oracle, bit for bit, and without a synchronous re-run beyond the first steps."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc = scenes.obb_pile(14, 8, 14, spacing=1.05)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    for i in range(70):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        assert g.counts() == o.counts(), f"step {i}"
    assert g.counts()["num_contacts"] > 3000
    assert g.physics_transforms()[0].tobytes() == o.physics_transforms()[0].tobytes()
    assert g.velocities()[0].tobytes() == o.velocities()[0].tobytes()
    tail_steps, tail_rounds = g.color_tail_stats()
    if env.get("MI_COLOR_TAIL") == "0": assert tail_steps == 0
    if "MI_COLOR_ROUNDS_MAX" in env: assert tail_steps >= 30 and tail_rounds >= 2 * tail_steps, "with one or two rounds enqueued the tail has to colour nearly everything"


@pytest.mark.parametrize("variant", ["0", "1"], ids=["GJK by lanes + EPA queue", "GJK and EPA by one wave per pair"])
@pytest.mark.parametrize("make", [lambda: scenes.shape_zoo(8, 5, 8), lambda: scenes.vehicles(3, 2), lambda: scenes.zones(8, 3, 8)], ids=["all shape pairs", "hull terrain", "triggers"])
def test_gpu_gjk_epa_variants_match_oracle(mi_lib, oracle_mod, monkeypatch, make, variant):
    """The two mappings of GJK / EPA to the chip (the library picks by the number of GJK pairs; MI_GJK_WAVE forces one): EPA always by
    one wave per pair with the polytope in LDS — closest face, visible faces, horizon edges and hull supports found by the lanes
    together, picking what the sequential scans pick — GJK either by one lane per pair or by the same wave.  Bit-identical to the
    oracle (which restates the reference's sequential EPA, and is pinned to the reference's own code)."""
    monkeypatch.setenv("MI_GJK_WAVE", variant)
    sc = make()
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    for i in range(160):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        assert g.counts() == o.counts(), f"step {i}"
        if i % 40 == 0:
            assert contact_set(g.contacts()) == contact_set(o.contacts()), f"step {i}"
    pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
    assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes()


@pytest.mark.parametrize("tail", ["0", "1"], ids=["margin of colouring rounds", "colouring tail"])
def test_gpu_speculative_step_retry_keeps_parity(mi_lib, oracle_mod, monkeypatch, tail):
    """Steps after the first run with ONE host read-back, sized from the previous step's counts.  Teleporting the bodies into
    a much denser pile invalidates those bounds: the step must be re-run synchronously from the untouched state and still match
    the oracle bit for bit.  (What this squeeze outgrows is the number of colouring rounds enqueued: with the colouring tail — the
    default — k_bin_hist runs the missing rounds itself and no re-run is needed; MI_COLOR_TAIL=0 keeps the re-run.)"""
    monkeypatch.setenv("MI_COLOR_TAIL", tail)
    sc = scenes.obb_pile(10, 4, 10, spacing=2.4)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    for _ in range(25):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
    steps, spec, retries0 = g.step_mode_stats()
    assert steps == 25 and spec >= 20
    ents = np.arange(sc.num_bodies, dtype=np.uint32)
    st = g.get_body_states(ents)
    assert st.tobytes() == o.get_body_states(ents).tobytes()
    st[:, 0] *= 0.42; st[:, 2] *= 0.42          # squeeze the lattice: several times more overlapping pairs at once
    g.set_body_states(ents, st); o.set_body_states(ents, st)
    for i in range(12):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        assert g.counts() == o.counts(), f"step {i}"
    if tail == "0": assert g.step_mode_stats()[2] > retries0, "the squeeze should have exceeded the speculative bounds"
    pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
    assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes()


def test_gpu_pair_partition_appearing_mid_run_keeps_parity(mi_lib, oracle_mod):
    """A box pile's pair list is of one type and is NOT partitioned by type; speculative steps then leave the partition pass out
    altogether.  The first cylinder that reaches a box populates a GJK bucket, so the list must be partitioned from that step
    on: the speculative step that left the pass out voids itself (k_pair_finish) and is re-run synchronously, and the
    trajectory stays the oracle's bit for bit across the switch and after it."""
    sc = scenes.obb_pile(8, 3, 8)
    n_new = 6
    e = scenes.make_entities(n_new)
    e["position"] = [(-3 + 1.3 * i, 7.5 + 0.4 * i, 0.7 * i - 2) for i in range(n_new)]
    e["rotation"][:, 3] = 1.0
    c = scenes.make_colliders(n_new, capi.CYLINDER)
    c["shape"][:, :7] = (0, -0.4, 0, 0, 0.4, 0, 0.35)
    ents = np.concatenate([sc.entities, e])
    first = len(sc.entities)
    sc = scenes.Scene("pile_then_cylinders", ents, np.concatenate([sc.collider_entities, np.arange(first, first + n_new, dtype=np.uint32)]),
                      np.concatenate([sc.colliders, c]), sc.solver_iterations)
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    for i in range(170):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        assert g.counts() == o.counts(), f"step {i}"
    pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
    assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes()
    assert pg[-n_new:, 1].max() < 6.0, "the cylinders were meant to land on the pile"
    steps, spec, retries = g.step_mode_stats()
    assert retries >= 1 and spec >= steps - 1 - 2 * retries


@pytest.mark.parametrize("name,make,steps", [
    ("cfg1", lambda: scenes.sphere_drop(16), 150),
    ("cfg2", lambda: scenes.mixed_stack(64, 16, 64), 60),
    ("cfg3", lambda: scenes.obb_pile(128, 16, 128), 60),
    ("cfg4", lambda: scenes.ragdolls(32, 32), 100),
    ("cfg5", lambda: scenes.vehicles(16, 16), 100),
])
def test_gpu_baseline_sizes_properties(mi_lib, name, make, steps):
    """The five BASELINE.json configurations at their FULL sizes (too slow for the oracle): size-independent properties.
    Two independent runs are bit-identical (the whole pipeline is deterministic although pairs, queues and tiles are filled
    in arrival order), the state stays finite with unit quaternions, nothing sinks through the ground / terrain, linear
    momentum is not created along x/z beyond what friction at the static ground allows (the pile's centre of mass stays put),
    the schedule colours are valid (contact counts consistent), and every step after the first ran speculatively."""
    sc = make()
    res = []
    for _ in range(2):
        w = sc.populate(gpu_world(mi_lib))
        s = sc.settings()
        p0, _ = w.physics_transforms()
        w.step_fixed(s, sc.dt, steps)
        p, q = w.physics_transforms(); v, a = w.velocities()
        res.append((p.tobytes(), q.tobytes(), v.tobytes(), a.tobytes(), w.counts()))
    assert res[0] == res[1]
    nb = sc.num_bodies
    assert np.isfinite(p).all() and np.isfinite(q).all() and np.isfinite(v).all() and np.isfinite(a).all()
    assert np.allclose(np.linalg.norm(q[:nb], axis=1), 1.0, atol=1e-4)
    floor = -0.8 if name == "cfg5" else 0.0          # cfg5: crowned terrain tiles with seeded +-0.15 m offsets, edges 0.7 m below the crown
    assert p[:nb, 1].min() > floor - 0.05
    c = w.counts()
    assert c["num_contacts"] >= c["num_collisions"] > 0 and c["num_contacts"] <= 4 * c["num_collisions"]
    assert c["num_colors"] <= 64
    if name in ("cfg1", "cfg2", "cfg3"):             # symmetric drops: the centre of mass does not wander sideways
        drift = np.abs(p[:nb, [0, 2]].mean(axis=0) - p0[:nb, [0, 2]].mean(axis=0)).max()
        assert drift < 0.05
    total, spec, retries = w.step_mode_stats()
    assert total == steps and spec >= steps - 1 - retries and retries <= 3


def _ragdolls_and_loose_boxes():
    """Ragdolls on the ground plus loose boxes dropped on some of them: islands that only touch the ground and themselves (private) next to
    islands coupled to a free body — and, once ragdolls are pushed into each other, to another island."""
    sc = scenes.ragdolls(4, 4)
    n_new = 6
    e = scenes.make_entities(n_new)
    first_body = sc.entities["position"][sc.entities["kind"] == capi.ENTITY_DYNAMIC]
    e["position"] = [first_body[i * 29 % len(first_body)] + np.array([0.1, 1.2 + 0.1 * i, 0.0], np.float32) for i in range(n_new)]
    e["rotation"][:, 3] = 1.0
    e["linear_velocity"][:, 0] = [1.5, -1.5, 0.0, 2.5, 0.0, -2.5]
    c = scenes.make_colliders(n_new, capi.AABB)
    c["shape"][:, :6] = (-0.25, -0.25, -0.25, 0.25, 0.25, 0.25)
    first = len(sc.entities)
    return scenes.Scene("ragdolls_and_boxes", np.concatenate([sc.entities, e]), np.concatenate([sc.collider_entities, np.arange(first, first + n_new, dtype=np.uint32)]),
                        np.concatenate([sc.colliders, c]), sc.solver_iterations, sc.dt, constraints=sc.constraints, global_constraints=sc.global_constraints)


@pytest.mark.parametrize("make,steps", [(lambda: scenes.ragdolls(5, 5), 150), (lambda: scenes.vehicles(4, 4), 120), (_ragdolls_and_loose_boxes, 150)],
                         ids=["ragdolls (cfg4): every island private", "vehicles on hull tiles (cfg5)", "ragdolls + loose boxes: private and coupled islands side by side"])
def test_gpu_private_islands_match_oracle_and_the_dataflow_path(mi_lib, oracle_mod, monkeypatch, make, steps):
    """An articulated island whose manifolds touch no dynamic body outside it (a ragdoll on the ground, a vehicle on static tiles) is solved by ONE
    workgroup for all sweeps — joints and contacts, bodies in LDS, a manifold's rows in a lane's registers (joints.hpp privateIsland) — instead of
    handing its bodies over to the contact tiles and back in every sweep.  Same canonical order restricted to the island: bit-identical to the oracle and
    to the world that sends every island through the dataflow (MI_ISLAND_PRIVATE=0), also while islands switch between the two treatments."""
    sc = make()
    monkeypatch.delenv("MI_ISLAND_PRIVATE", raising=False)
    g = sc.populate(gpu_world(mi_lib))
    monkeypatch.setenv("MI_ISLAND_PRIVATE", "0")
    d = sc.populate(gpu_world(mi_lib))
    monkeypatch.delenv("MI_ISLAND_PRIVATE", raising=False)
    o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    for i in range(steps):
        g.step_fixed(s, sc.dt, 1); d.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        assert g.counts() == o.counts() == d.counts(), f"step {i}"
        if i % 10 == 9 or i == steps - 1:
            pg, qg = g.physics_transforms(); po, qo = o.physics_transforms(); pd, qd = d.physics_transforms()
            assert pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes(), f"step {i}: private islands vs oracle"
            assert pd.tobytes() == po.tobytes() and qd.tobytes() == qo.tobytes(), f"step {i}: dataflow islands vs oracle"
    vg, wg = g.velocities(); vo, wo = o.velocities()
    assert vg.tobytes() == vo.tobytes() and wg.tobytes() == wo.tobytes()
    assert g.solver_kind() == 3 and d.solver_kind() == 3
    total, spec, retries = g.step_mode_stats()
    assert spec >= total - 2 - retries


def test_gpu_bench_size_solvers_agree(mi_lib, monkeypatch):
    """BASELINE's 262 144-body pile, far beyond the oracle's reach: the default solver (XCD-partitioned persistent kernel: eight tile
    lists, ~95 % of the bodies handed over through an XCD's L2, the seam bodies through memory) must end bit-identical to the
    dispatch-ordered flow kernel, to the unpartitioned persistent kernel — which the small cases pin to the oracle — and to a world
    that takes every step synchronously (MI_ASYNC=0: exact sizes read back inside the step; the path of every re-run) — and to worlds stepping with round 4's
    launches, or with the colouring rounds moved into the tail."""
    import hashlib
    sc = scenes.obb_pile(128, 16, 128)
    out = {}
    round4_step = {"MI_FUSE_RESET": "0", "MI_ROUND0_EMIT": "0", "MI_FUSE_LARGE": "0", "MI_FINISH_IN_NARROW": "0", "MI_COLOR_TAIL": "0", "MI_FUSE_KEYS": "0"}   # every launch round 5 removed, back
    for name, env in (("default", {}), ("flow", {"MI_SOLVER": "flow"}), ("unpartitioned", {"MI_PERSIST_XCD": "0"}), ("synchronous", {"MI_ASYNC": "0"}),
                      ("the tail colours", {"MI_COLOR_ROUNDS_MAX": "1"}), ("29 launches", round4_step), ("every tile's rows stream", {"MI_PERSIST_RESIDENT": "0"}), ("no step-ahead", {"MI_STEP_AHEAD": "0"})):
        for k in ("MI_SOLVER", "MI_PERSIST_XCD", "MI_ASYNC", "MI_COLOR_ROUNDS_MAX", "MI_PERSIST_RESIDENT", "MI_STEP_AHEAD", *round4_step):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        w = sc.populate(gpu_world(mi_lib))
        w.step_fixed(sc.settings(), sc.dt, 120)
        p, q = w.physics_transforms()
        out[name] = (hashlib.sha1(p.tobytes() + q.tobytes()).hexdigest(), w.counts()["num_contacts"], w.solver_kind(), w.step_mode_stats()[2])
        w.close()
    assert out["default"][2] == 4 and out["flow"][2] == 1 and out["unpartitioned"][2] == 2, out
    assert out["default"][:2] == out["flow"][:2] == out["unpartitioned"][:2] == out["synchronous"][:2], out   # (synchronous: every step sized from read-backs inside the step — the path every re-run takes)
    # the launches a steady step no longer makes (knobs.hpp) change nothing either; nor does it matter who runs the colouring rounds (MI_COLOR_ROUNDS_MAX=1: all but
    # one of them inside k_bin_hist, 256 workgroups striding over 360 k manifolds with a device-wide barrier in between)
    assert out["the tail colours"][:2] == out["default"][:2] and out["29 launches"][:2] == out["default"][:2], out
    assert out["default"][1] > 150000
    assert out["default"][3] <= 12, "speculative retries while the pile lands are fine; a solver that keeps falling back is not"


def test_gpu_full_size_properties(mi_lib):
    """BASELINE sizes are too slow for the oracle; check size-independent properties instead:
    determinism (two runs bit-identical), no body below the ground, finite state, valid colouring."""
    sc = scenes.obb_pile(64, 8, 64)
    res = []
    for _ in range(2):
        w = sc.populate(gpu_world(mi_lib))
        w.step_fixed(sc.settings(), sc.dt, 40)
        p, q = w.physics_transforms()
        res.append((p.tobytes(), q.tobytes(), w.counts()))
    assert res[0] == res[1]
    assert np.isfinite(p).all() and np.isfinite(q).all()
    nb = sc.num_bodies
    assert p[:nb, 1].min() > 0.0
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-4)
    # colouring: no two manifolds of one colour share a dynamic body
    c = w.contacts()
    L = w.L
    import ctypes as C
    nm = w.counts()["num_collisions"]
    colors = np.zeros(nm, np.uint32)
    L.check(L.fn("world_get_manifold_colors")(w.h, colors.ctypes.data_as(C.c_void_p), C.c_uint32(nm)), "colors")
    first = np.r_[True, (c["collider_a"][1:] != c["collider_a"][:-1]) | (c["collider_b"][1:] != c["collider_b"][:-1])]
    ba, bb = c["body_a"][first], c["body_b"][first]
    assert len(ba) == nm
    for col in np.unique(colors):
        sel = colors == col
        bodies = np.concatenate([ba[sel], bb[sel]])
        bodies = bodies[bodies < nb]
        assert len(np.unique(bodies)) == len(bodies), f"colour {col} reuses a body"


def test_gpu_body_state_exchange_api(mi_lib):
    """The ghost-exchange entry points: host variant vs device-pointer variant (torch CUDA tensors, as used with RCCL)."""
    import torch
    sc = scenes.obb_pile(6, 3, 6)
    w = sc.populate(gpu_world(mi_lib))
    w.step_fixed(sc.settings(), sc.dt, 5)
    ents = np.arange(10, 60, dtype=np.uint32)
    host = w.get_body_states(ents)
    p, q = w.physics_transforms(); v, a = w.velocities()
    assert np.array_equal(host[:, 0:3], p[ents]) and np.array_equal(host[:, 3:7], q[ents])
    assert np.array_equal(host[:, 7:10], v[ents]) and np.array_equal(host[:, 10:13], a[ents])
    ids = torch.from_numpy(w.entities_to_bodies(ents).astype(np.int32)).cuda()
    buf = torch.zeros(len(ents) * 13, dtype=torch.float32, device="cuda")
    w.get_body_states_device(len(ents), ids.data_ptr(), buf.data_ptr())
    assert np.array_equal(buf.cpu().numpy().reshape(-1, 13), host)
    buf2 = buf.clone(); buf2.view(-1, 13)[:, 1] += 2.0          # lift them 2 m
    torch.cuda.synchronize()
    w.set_body_states_device(len(ents), ids.data_ptr(), buf2.data_ptr())
    after = w.get_body_states(ents)
    assert np.allclose(after[:, 1], host[:, 1] + 2.0) and np.array_equal(after[:, 3:], host[:, 3:])
    w.step_fixed(sc.settings(), sc.dt, 2)                        # keeps stepping from the new state
    assert np.isfinite(w.physics_transforms()[0]).all()


# (the sharded-world tests live in tests/test_gpu_sharding.py)


@pytest.mark.gpu
@pytest.mark.parametrize("make,victims", [(lambda: scenes.shape_zoo(), [3, 77, 143, 10, 11, 142]), (lambda: scenes.ragdolls(3, 3), [5, 20, 55, 0, 100]),
                                          (lambda: scenes.zones(), [4, 111, 60, 113, 108])], ids=["zoo", "ragdolls", "zones"])
def test_gpu_entity_deletion_matches_oracle(mi_lib, oracle_mod, make, victims):
    """mi_entity_destroy = game_scene::deleteEntity (scene.cpp:124-150) while the simulation runs: bodies with one and several
    colliders, bodies with joints, triggers, a force field.  The pools follow EnTT's swap-and-pop (the oracle's deletion is pinned
    to the reference's own, tests/test_reference_pin.py), events keep flowing, everything stays bit-identical."""
    sc = make()
    g = sc.populate(gpu_world(mi_lib)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    for w in (g, o):
        w.enable_events(True)
    s = sc.settings()
    schedule = {15 + 12 * k: v for k, v in enumerate(victims)}
    for i in range(120):
        if i in schedule:
            for w in (g, o):
                w.destroy_entity(schedule[i])
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        assert g.counts() == o.counts(), f"step {i}"
        assert g.poll_events().tobytes() == o.poll_events().tobytes(), f"step {i}: events"
    for a, b in zip(g.physics_transforms() + g.velocities() + g.transforms(), o.physics_transforms() + o.velocities() + o.transforms()):
        assert a.tobytes() == b.tobytes()
    with pytest.raises(mi_lib.PhysicsError):
        g.destroy_entity(victims[0])


def test_gpu_step_ahead_is_adopted_or_dropped_and_never_seen(mi_lib, oracle_mod, monkeypatch):
    """A speculative step enqueues the next step's first kernel (k_bp_prepare) behind its own end-of-step record (csrc/world.hip, "Step-ahead").  Free-running, the next step
    adopts it; a state write between two steps (mi_world_set_body_states), an entity deletion (re-upload) and a step that is re-run make the next step drop it — and in
    every case the world is, bit for bit, the oracle's and the world that never runs anything ahead (MI_STEP_AHEAD=0)."""
    sc = scenes.obb_pile(24, 6, 24, spacing=1.0)         # 3 456 boxes: above the step-graph limit? no — graphs are off for this test (a replayed graph holds its own first kernel)
    monkeypatch.setenv("MI_GRAPH", "0")
    monkeypatch.setenv("MI_STEP_AHEAD", "1")             # the default, stated: the suite is also run with the knob off
    a = sc.populate(gpu_world(mi_lib))
    monkeypatch.setenv("MI_STEP_AHEAD", "0")
    b = sc.populate(gpu_world(mi_lib))
    o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings(); ids = np.arange(sc.num_bodies, dtype=np.uint32)
    for i in range(90):
        for w in (a, b, o):
            w.step_fixed(s, sc.dt, 1)
        assert a.counts() == o.counts() == b.counts(), f"step {i}"
        if i % 10 == 9:
            assert a.get_body_states(ids).tobytes() == o.get_body_states(ids).tobytes() == b.get_body_states(ids).tobytes(), f"step {i}"
        if i == 40:      # an outside write: four boxes are thrown upwards
            st = a.get_body_states(ids[:4]); st[:, 7:10] = (0.0, 9.0, 0.0)
            for w in (a, b, o):
                w.set_body_states(ids[:4], st)
        if i == 60:      # a topology edit: everything is uploaded again
            for w in (a, b, o):
                w.destroy_entity(5)
            ids = np.asarray([e for e in range(sc.num_bodies) if e != 5], dtype=np.uint32)
    enq, adopted = a.debug_step_ahead_stats()
    assert enq >= 60 and 0 < adopted < enq, (enq, adopted)                # adopted in the free-running stretches, dropped after the write / the upload / a re-run
    assert b.debug_step_ahead_stats() == (0, 0)
