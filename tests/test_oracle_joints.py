"""Known-answer tests for the oracle's joint constraints (the reference has none of its own)."""
import math
import numpy as np
import pytest

from d3d12renderer_amd import capi, scenes

DT = 1.0 / 120.0


def two_boxes(kind_a=capi.ENTITY_KINEMATIC, pos_b=(1.0, 5.0, 0.0), damping=0.0, half=(0.1, 0.1, 0.1)):
    e = scenes.make_entities(2)
    e["position"][0] = (0, 5, 0)
    e["position"][1] = pos_b
    e["kind"][0] = kind_a
    e["linear_damping"] = damping
    e["angular_damping"] = damping
    c = scenes.make_colliders(2, capi.AABB, density=10.0)
    c["shape"][:, :6] = (-half[0], -half[1], -half[2], half[0], half[1], half[2])
    return e, c


def build(oracle_mod, e, c, joints, order=0, iterations=30):
    sc = scenes.Scene("j", e, np.arange(len(e), dtype=np.uint32), c, iterations, global_constraints=joints)
    return sc, sc.populate(oracle_mod.create_world(order))


def quat_rotate(q, v):
    return scenes.q_rot(np.asarray(q, np.float64), np.asarray(v, np.float64))


def test_distance_constraint_keeps_length(oracle_mod):
    e, c = two_boxes(damping=0.8)
    sc, w = build(oracle_mod, e, c, [(capi.CONSTRAINT_DISTANCE, 0, 1, (0, 5, 0), (1, 5, 0), 0, 0, {})])
    w.step_fixed(sc.settings(), DT, 1500)
    p, _ = w.physics_transforms()
    assert abs(np.linalg.norm(p[1] - p[0]) - 1.0) < 2e-2
    assert p[1, 1] < 4.1 and abs(p[1, 0]) < 0.1   # damped: it ends hanging straight down


def test_ball_pendulum_period(oracle_mod):
    # point-like bob on a ball joint: small-angle period 2 pi sqrt(L/g); undamped
    L = 2.0
    th0 = 0.1
    e, c = two_boxes(pos_b=(L * math.sin(th0), 5.0 - L * math.cos(th0), 0.0), half=(0.02, 0.02, 0.02))
    sc, w = build(oracle_mod, e, c, [(capi.CONSTRAINT_BALL, 0, 1, (0, 5, 0), None, 0, 0, {})])
    s = sc.settings()
    xs = []
    for _ in range(720):
        w.step_fixed(s, DT, 1)
        xs.append(w.physics_transforms()[0][1, 0])
    xs = np.asarray(xs)
    zc = np.where((xs[:-1] > 0) & (xs[1:] <= 0))[0]
    assert len(zc) >= 2
    period = (zc[1] - zc[0]) * DT
    assert period == pytest.approx(2 * math.pi * math.sqrt(L / 9.81), rel=0.05)
    p = w.physics_transforms()[0]
    assert abs(np.linalg.norm(p[1] - p[0]) - L) < 2e-2


def test_fixed_constraint_holds_relative_pose(oracle_mod):
    e, c = two_boxes(pos_b=(0.5, 5.0, 0.0))
    sc, w = build(oracle_mod, e, c, [(capi.CONSTRAINT_FIXED, 0, 1, (0.25, 5, 0), None, 0, 0, {})])
    w.step_fixed(sc.settings(), DT, 300)
    p, q = w.physics_transforms()
    assert np.allclose(p[1] - p[0], (0.5, 0, 0), atol=2e-2)
    assert abs(abs(q[1, 3]) - 1.0) < 1e-3


def test_hinge_keeps_axes_aligned_and_respects_limits(oracle_mod):
    e, c = two_boxes(pos_b=(0.6, 5.0, 0.0), half=(0.25, 0.05, 0.05))
    sc, w = build(oracle_mod, e, c, [(capi.CONSTRAINT_HINGE, 0, 1, (0.3, 5, 0), (0, 0, 1), -0.5, 0.2, {})])
    s = sc.settings()
    angles = []
    for _ in range(400):
        w.step_fixed(s, DT, 1)
        _, q = w.physics_transforms()
        zb = quat_rotate(q[1], (0, 0, 1))
        assert zb[2] > 0.999                      # only rotates about z
        xb = quat_rotate(q[1], (1, 0, 0))
        angles.append(math.atan2(xb[1], xb[0]))
    assert min(angles) > -0.5 - 0.05              # gravity swings it down to the min limit and it stays there
    assert min(angles) < -0.4
    assert max(angles) <= 0.2 + 0.05


def test_hinge_velocity_motor_reaches_target_speed(oracle_mod):
    e, c = two_boxes(pos_b=(0.0, 5.0, 0.5), half=(0.2, 0.2, 0.05))
    e["gravity_factor"] = 0.0
    sc, w = build(oracle_mod, e, c, [(capi.CONSTRAINT_HINGE, 0, 1, (0, 5, 0.25), (0, 0, 1), 1.0, -1.0,
                                     {"max_motor_torque": 50.0, "motor_type": 0, "motor_velocity_or_target_angle": 2.0})])
    w.step_fixed(sc.settings(), DT, 240)
    _, ang = w.velocities()
    assert ang[1, 2] == pytest.approx(2.0, abs=0.05)


def test_slider_moves_only_along_axis_within_limits(oracle_mod):
    e, c = two_boxes(pos_b=(0.0, 4.5, 0.0))
    sc, w = build(oracle_mod, e, c, [(capi.CONSTRAINT_SLIDER, 0, 1, (0, 4.75, 0), (0, 1, 0), -0.4, 0.1, {})])
    w.step_fixed(sc.settings(), DT, 400)
    p, q = w.physics_transforms()
    d = p[1] - np.array([0, 4.5, 0])
    assert abs(d[0]) < 1e-2 and abs(d[2]) < 1e-2
    assert -0.4 - 0.03 < d[1] < -0.35             # slid down to the negative limit
    assert abs(abs(q[1, 3]) - 1.0) < 1e-3


def test_cone_twist_swing_limit(oracle_mod):
    e, c = two_boxes(pos_b=(0.6, 5.0, 0.0), half=(0.25, 0.05, 0.05))
    lim = 0.4
    sc, w = build(oracle_mod, e, c, [(capi.CONSTRAINT_CONE_TWIST, 0, 1, (0.3, 5, 0), (1, 0, 0), lim, 0.3, {})])
    s = sc.settings()
    worst = 0.0
    for _ in range(400):
        w.step_fixed(s, DT, 1)
        _, q = w.physics_transforms()
        xb = quat_rotate(q[1], (1, 0, 0))
        worst = max(worst, math.acos(max(-1.0, min(1.0, xb[0]))))
    assert lim - 0.05 < worst < lim + 0.08


def test_from_global_matches_template_pods(oracle_mod):
    """add*ConstraintFromGlobalPoints at the ragdoll's base pose must reproduce the PODs the scene generator computes."""
    pos, rot, joints = scenes._ragdoll_template()
    n = len(pos)
    e = scenes.make_entities(n)
    for i in range(n):
        e["position"][i] = pos[i]; e["rotation"][i] = rot[i]
    c = scenes.make_colliders(n, capi.SPHERE); c["shape"][:, 3] = 0.1
    w = oracle_mod.create_world(0)
    w.create_entities(e); w.add_colliders(np.arange(n, dtype=np.uint32), c)
    sc = scenes._RAGDOLL_SCALE
    for (kind, a, b, ap, la, ax, l0, l1), (ctype, ia, ib, pod) in zip(scenes._RAGDOLL_JOINTS, joints):
        ipn = scenes._P[ap]
        anchor = scenes.q_rot(rot[ipn], sc * np.asarray(la, np.float64)) + pos[ipn]
        axis = scenes.q_rot(rot[scenes._P[ax[1]]], np.asarray(ax[2], np.float64)) if isinstance(ax[0], str) else np.asarray(ax, np.float64)
        lim0 = -1.0 if l0 is None else (np.deg2rad(l0))
        cid = w.add_constraint_from_global(ctype, ia, ib, anchor.astype(np.float32), axis.astype(np.float32), float(lim0), float(np.deg2rad(l1)))
        got = w.get_constraint(ctype, cid)
        for name in pod.dtype.names:
            assert np.allclose(got[name], pod[name], atol=2e-6), (kind, a, b, name)


def test_ragdoll_mass_and_joint_counts(oracle_mod):
    sc = scenes.ragdolls(2, 2)
    assert sc.num_bodies == 4 * 14 and len(sc.colliders) == 4 * 17 + 1 and len(sc.constraints) == 4 * 13
    w = sc.populate(oracle_mod.create_world(0))
    im, _, _ = w.mass_properties()
    assert 60.0 < (1.0 / im[:14]).sum() < 120.0   # a human (density 985 kg/m^3)
    w.step_fixed(sc.settings(), sc.dt, 200)
    p, _ = w.physics_transforms()
    assert np.isfinite(p).all() and p[:-1, 1].min() > -0.05


def test_constraint_deletion_semantics(oracle_mod):
    """deleteConstraint / deleteAllConstraintsFromEntity / deleteAllConstraints (physics.cpp:443-539): handles of the other
    constraints stay valid (EnTT swap-and-pop only reorders the pool), a released body falls freely."""
    from d3d12renderer_amd import scenes
    sc = scenes.ragdolls(1, 1)
    w = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_REFERENCE))
    before = [w.get_constraint(capi.CONSTRAINT_HINGE, i).copy() for i in range(6)]
    w.destroy_constraint(capi.CONSTRAINT_HINGE, 1)
    with pytest.raises(capi.PhysicsError):
        w.get_constraint(capi.CONSTRAINT_HINGE, 1)
    with pytest.raises(capi.PhysicsError):
        w.destroy_constraint(capi.CONSTRAINT_HINGE, 1)
    for i in (0, 2, 3, 4, 5):
        assert w.get_constraint(capi.CONSTRAINT_HINGE, i).tobytes() == before[i].tobytes()
    pod = w.get_constraint(capi.CONSTRAINT_HINGE, 5); pod["max_motor_torque"] = 77.0
    w.update_constraint(capi.CONSTRAINT_HINGE, 5, pod)
    assert w.get_constraint(capi.CONSTRAINT_HINGE, 5)["max_motor_torque"][0] == 77.0
    new_id = w.add_constraint(capi.CONSTRAINT_HINGE, 2, 3, before[1])       # ids are never reused
    assert new_id == 6
    # the head (entity 1) hangs on the neck cone-twist only: release it and it is a free body
    w.destroy_entity_constraints(1)
    with pytest.raises(capi.PhysicsError):
        w.get_constraint(capi.CONSTRAINT_CONE_TWIST, 0)
    w.get_constraint(capi.CONSTRAINT_CONE_TWIST, 1)
    w.step_fixed(sc.settings(), sc.dt, 1)
    w.destroy_all_constraints()
    for t, n in ((capi.CONSTRAINT_HINGE, 7), (capi.CONSTRAINT_CONE_TWIST, 7)):
        for i in range(n):
            with pytest.raises(capi.PhysicsError):
                w.get_constraint(t, i)
    w.step_fixed(sc.settings(), sc.dt, 30)
    assert np.isfinite(w.physics_transforms()[0]).all()
