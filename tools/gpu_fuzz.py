"""Differential fuzzer: random small worlds stepped on the GPU (the product library through the C-ABI) and in the oracle (CPU restatement, canonical order), compared bit for bit.

    python tools/gpu_fuzz.py --seeds 0:400 --budget 600 [--steps 40] [--out gpurun_out/fuzz.json]

A world of one seed: 6..160 entities in a jittered lattice whose spacing ranges from overlapping to loose, over a ground box (sometimes a tilted OBB); dynamic, kinematic and static
entities; one to three colliders per entity of every type (sphere, capsule, cylinder, AABB, OBB, convex hull), sizes from a few centimetres to colliders large enough for the
broad phase's large-collider pass; materials, damping, gravity factors, initial velocities at random; joints of every type between lattice neighbours in half of the worlds; 1..30
solver iterations; a rolling heightmap instead of the ground box in a fifth of the worlds; triggers and force fields (with and without colliders) in a quarter; collision
and trigger events on in half.  Every world is stepped once in the oracle (recorded) and then on the GPU under the default environment AND under one of the library's other
paths, chosen by the seed (step graphs forced, XCD-partitioned persistent solver on a small pile, synchronous steps, the dispatch-ordered and the per-colour solver, both GJK
variants, no step-ahead ...): all of them must reproduce the recording.  While stepping: velocity kicks written through mi_world_set_body_states, entities deleted (re-upload), forces applied,
the state saved and restored in place, joints removed, new bodies dropped in; a quarter of the worlds take uneven frame times through physicsStep's accumulator.  Compared every step: the counts
(bodies, overlaps, collisions, contacts); every fifth step and at the end: the 13 floats of every body's state, as bytes.  The oracle is the checker here — test infrastructure,
like tests/test_gpu_parity.py, of which this is the randomised sibling (the reference has no such test; its physics is checked by eye in the editor)."""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import d3d12renderer_amd as mi                      # noqa: E402
from d3d12renderer_amd import capi, scenes          # noqa: E402

JOINT_FLAVOURS = [
    (capi.CONSTRAINT_DISTANCE, None, 1.0, -1.0, {}),
    (capi.CONSTRAINT_BALL, None, 1.0, -1.0, {}),
    (capi.CONSTRAINT_FIXED, None, 1.0, -1.0, {}),
    (capi.CONSTRAINT_HINGE, (0, 0, 1), 1.0, -1.0, {}),
    (capi.CONSTRAINT_HINGE, (0, 0, 1), -0.3, 0.4, {}),
    (capi.CONSTRAINT_HINGE, (0, 1, 0), 1.0, -1.0, {"max_motor_torque": 40.0, "motor_type": 0, "motor_velocity_or_target_angle": 1.5}),
    (capi.CONSTRAINT_HINGE, (1, 0, 0), -1.0, 1.0, {"max_motor_torque": 60.0, "motor_type": 1, "motor_velocity_or_target_angle": 0.6}),
    (capi.CONSTRAINT_CONE_TWIST, (1, 0, 0), 0.5, 0.3, {}),
    (capi.CONSTRAINT_CONE_TWIST, (0, 1, 0), 0.9, 0.6, {"max_swing_motor_torque": 30.0, "swing_motor_type": 1, "swing_motor_velocity_or_target_angle": 0.4,
                                                     "swing_motor_axis": 0.7, "max_twist_motor_torque": 20.0, "twist_motor_type": 0, "twist_motor_velocity_or_target_angle": 1.0}),
    (capi.CONSTRAINT_SLIDER, (1, 0, 0), 1.0, -1.0, {}),
    (capi.CONSTRAINT_SLIDER, (0, 1, 0), -0.2, 0.3, {}),
    (capi.CONSTRAINT_SLIDER, (0, 0, 1), -0.5, 0.5, {"max_motor_force": 200.0, "motor_type": 1, "motor_velocity_or_target_distance": 0.25}),
]


def random_collider(rng, hull_count, big):
    k = int(rng.integers(0, 6))
    r = float(rng.uniform(0.08, 0.55)) * (rng.uniform(3.0, 8.0) if big else 1.0)
    h = float(rng.uniform(0.05, 0.6)) * (rng.uniform(2.0, 5.0) if big else 1.0)
    off = tuple(float(x) for x in rng.uniform(-0.3, 0.3, 3)) if rng.random() < 0.3 else (0.0, 0.0, 0.0)
    mat = {"restitution": float(rng.uniform(0.0, 1.0)), "friction": float(rng.uniform(0.0, 1.2)), "density": float(rng.uniform(0.3, 6.0))}
    if k == capi.SPHERE:
        return (k, (*off, r), mat)
    if k in (capi.CAPSULE, capi.CYLINDER):
        d = rng.normal(size=3); d /= max(1e-6, np.linalg.norm(d)); d *= h
        return (k, (off[0] - d[0], off[1] - d[1], off[2] - d[2], off[0] + d[0], off[1] + d[1], off[2] + d[2], r * 0.7), mat)
    if k == capi.AABB:
        return (k, (off[0] - r, off[1] - h, off[2] - r * 0.8, off[0] + r, off[1] + h, off[2] + r * 0.8), mat)
    if k == capi.OBB:
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        return (k, (float(q[0]), float(q[1]), float(q[2]), float(q[3]), *off, r, h, r * float(rng.uniform(0.5, 1.5))), mat)      # {q, centre, radii} (include/mi_physics.h)
    q = rng.normal(size=4) if rng.random() < 0.5 else np.array([0.0, 0.0, 0.0, 1.0]); q /= np.linalg.norm(q)
    return (k, (float(q[0]), float(q[1]), float(q[2]), float(q[3]), *off), dict(mat, hull=int(rng.integers(0, hull_count))))             # {q, position}


def make_world_description(seed, scale=1):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(6, 161)) * scale
    side = max(1, int(round((n / rng.uniform(1.0, 6.0)) ** 0.5)))
    spacing = float(rng.uniform(0.45, 1.6))
    jitter = float(rng.uniform(0.0, 0.3))
    hulls = [scenes.convex_hull_mesh(seed * 7 + 1 + i, n_points=int(rng.integers(6, 40)), radius=float(rng.uniform(0.25, 0.7))) for i in range(int(rng.integers(1, 4)))]
    ents, cent, cols, col_hull = [], [], [], []
    kinds = []
    for i in range(n):
        u = rng.random()
        kind = capi.ENTITY_DYNAMIC if u < 0.88 else capi.ENTITY_KINEMATIC if u < 0.94 else capi.ENTITY_STATIC
        kinds.append(kind)
        e = scenes.make_entities(1, kind)
        ix, iz, iy = i % side, (i // side) % side, i // (side * side)
        e["position"][0] = (np.array([(ix - side / 2) * spacing, 0.4 + iy * spacing, (iz - side / 2) * spacing]) + rng.uniform(-jitter, jitter, 3)).astype(np.float32)
        q = rng.normal(size=4) if rng.random() < 0.8 else np.array([0.0, 0.0, 0.0, 1.0]); q /= np.linalg.norm(q)
        e["rotation"][0] = q.astype(np.float32)
        if kind != capi.ENTITY_STATIC and rng.random() < 0.5:
            e["linear_velocity"][0] = rng.uniform(-2.0, 2.0, 3).astype(np.float32)
            e["angular_velocity"][0] = rng.uniform(-4.0, 4.0, 3).astype(np.float32)
        e["gravity_factor"][0] = float(rng.choice([1.0, 1.0, 1.0, 0.0, 0.5, 2.0]))
        e["linear_damping"][0] = float(rng.choice([0.0, 0.4, rng.uniform(0.0, 2.0)]))
        e["angular_damping"][0] = float(rng.choice([0.0, 0.4, rng.uniform(0.0, 2.0)]))
        ents.append(e)
        for _ in range(1 if rng.random() < 0.85 else int(rng.integers(2, 4))):
            ctype, shape, mat = random_collider(rng, len(hulls), big=rng.random() < 0.03)
            c = scenes.make_colliders(1, ctype, mat["restitution"], mat["friction"], mat["density"])
            c["shape"][0, :len(shape)] = shape
            if ctype == capi.HULL:
                c["hull_geometry"][0] = mat["hull"]
            cols.append(c); cent.append(i)
    # triggers and force fields (handleNonCollisionInteractions): entities of their own, some without colliders (a force field without colliders is global)
    forces = []
    if rng.random() < 0.25:
        for _ in range(int(rng.integers(1, 4))):
            zi = len(ents)
            z = scenes.make_entities(1, capi.ENTITY_FORCE_FIELD if rng.random() < 0.5 else capi.ENTITY_TRIGGER)
            z["position"][0] = (rng.uniform(-1.0, 1.0, 3) * np.array([side * spacing / 2, 1.0, side * spacing / 2]) + np.array([0.0, 1.0, 0.0])).astype(np.float32)
            q = rng.normal(size=4); q /= np.linalg.norm(q); z["rotation"][0] = q.astype(np.float32)
            ents.append(z)
            if z["kind"][0] == capi.ENTITY_FORCE_FIELD:
                forces.append((zi, tuple(float(x) for x in rng.uniform(-12.0, 12.0, 3))))
            for _ in range(int(rng.integers(0, 3)) if z["kind"][0] == capi.ENTITY_FORCE_FIELD else int(rng.integers(1, 3))):
                ctype, shape, mat = random_collider(rng, len(hulls), big=False)
                if ctype == capi.HULL:
                    ctype, shape = capi.SPHERE, (0.0, 0.0, 0.0, 0.8)
                c = scenes.make_colliders(1, ctype); c["shape"][0, :len(shape)] = np.asarray(shape, np.float32) * (1.0 if ctype == capi.OBB else 2.5)
                if ctype == capi.OBB:
                    c["shape"][0, 7:10] *= 2.5
                cols.append(c); cent.append(zi)
    # the ground: a box (sometimes tilted), or rolling terrain
    heightmap = None
    if rng.random() < 0.2:
        amp = float(rng.uniform(1.0, 5.0)); size = float(rng.uniform(10.0, 24.0))
        heightmap = scenes.rolling_heightmap(chunks_per_dim=1, chunk_size=size, amplitude=amp, seed=seed + 1000)
        lift = amp * 0.62 + 0.3
        for e in ents:
            if e["kind"][0] not in (capi.ENTITY_FORCE_FIELD, capi.ENTITY_TRIGGER):
                e["position"][0, 1] += lift
    g = scenes.make_entities(1, capi.ENTITY_STATIC)
    gc = scenes.make_colliders(1, capi.AABB, 0.1, float(rng.uniform(0.2, 1.0)), 4.0)
    if rng.random() < 0.25:
        gc["type"][0] = capi.OBB
        q = np.array([rng.uniform(-0.05, 0.05), 0.0, rng.uniform(-0.05, 0.05), 1.0]); q /= np.linalg.norm(q)
        gc["shape"][0, :10] = (*q, 0, -2.0, 0, 60.0, 2.0, 60.0)
    else:
        gc["shape"][0, :6] = (-60.0, -4.0, -60.0, 60.0, 0.0, 60.0)
    if heightmap is None:
        cent.append(len(ents)); ents.append(g); cols.append(gc)
    E = np.concatenate(ents); Cd = np.concatenate(cols)
    gcs = []
    if rng.random() < 0.5:
        dyn = [i for i in range(n) if kinds[i] != capi.ENTITY_STATIC]
        for _ in range(int(rng.integers(1, max(2, n // 6)))):
            if len(dyn) < 2:
                break
            a = int(rng.choice(dyn)); b = a + int(rng.choice([1, side, side * side]))
            if b >= n or kinds[b] == capi.ENTITY_STATIC or (kinds[a] == capi.ENTITY_KINEMATIC and kinds[b] == capi.ENTITY_KINEMATIC):
                continue
            ctype, axis, l0, l1, edits = JOINT_FLAVOURS[int(rng.integers(0, len(JOINT_FLAVOURS)))]
            pa, pb = E["position"][a].copy(), E["position"][b].copy()
            if ctype == capi.CONSTRAINT_DISTANCE:
                gcs.append((ctype, a, b, pa, pb, l0, l1, edits))
            else:
                gcs.append((ctype, a, b, ((pa + pb) * 0.5).astype(np.float32), None if axis is None else np.asarray(axis, np.float32), l0, l1, edits))
    iters = int(rng.choice([1, 4, 10, 20, 30]))
    sc = scenes.Scene(f"fuzz_{seed}", E, np.asarray(cent, np.uint32), Cd, iters, hulls=hulls, global_constraints=gcs, forces=forces, heightmap=heightmap)
    sc.dt = float(rng.choice([1.0 / 120.0, 1.0 / 60.0, 1.0 / 240.0]))
    bodies = np.asarray([i for i in range(n) if kinds[i] != capi.ENTITY_STATIC], np.uint32)
    return sc, bodies, rng


# environments the GPU world is created under besides the default one (csrc/knobs.hpp: read per world): one of them per seed
PROGRESS = None
EXPLODED = []
TIMES = []
ENVIRONMENTS = [
    {"MI_GRAPH": "force"}, {"MI_GRAPH": "0"}, {"MI_ASYNC": "0"}, {"MI_SOLVER": "flow"}, {"MI_SOLVER": "launch"}, {"MI_SOLVER": "persist-global"}, {"MI_SOLVER": "persist-granules"},
    {"MI_PERSIST_XCD_MIN": "1", "MI_PERSIST_WAVES": "64"}, {"MI_PERSIST_XCD_MIN": "1", "MI_PERSIST_XCD_SINGLE": "0"}, {"MI_PERSIST_WAVES": "8", "MI_PERSIST_XCD": "0"},
    {"MI_GJK_WAVE": "0"}, {"MI_GJK_WAVE": "1"}, {"MI_STEP_AHEAD": "0"}, {"MI_PERSIST_RESIDENT": "0"}, {"MI_COLOR_TAIL": "0"}, {"MI_COLOR_ROUNDS_MAX": "1"}, {"MI_ROUND0_EMIT": "0"},
    {"MI_FUSE_WORLD": "0", "MI_FUSE_LARGE": "0", "MI_FUSE_KEYS": "0", "MI_FINISH_IN_NARROW": "0", "MI_FUSE_RESET": "0"}, {"MI_READBACK": "copy"}, {"MI_POSE_STREAM": "0"}, {"MI_ISLAND_PRIVATE": "0"},
]


def plan_actions(seed, steps):
    """What happens between the steps and how each step is taken — drawn once per seed as raw random numbers, resolved against the list of live bodies at run time, the same
    way in every world."""
    rng = np.random.default_rng(seed ^ 0x5EED)
    frames = rng.random() < 0.25          # a quarter of the worlds are stepped through physicsStep's accumulator (mi_world_step) with uneven frame times
    plan = []
    for i in range(steps):
        mode = ("frame", float(rng.uniform(0.002, 0.03))) if frames else ("fixed",)
        plan.append((mode, float(rng.random()), rng.random(8), rng.uniform(-1.0, 1.0, (4, 3)).astype(np.float32)))
    return plan


def apply_action(w, u, r, v, alive, sc):
    """One action between two steps (u picks it; r, v are its random payload).  Returns the new list of live bodies."""
    pick = lambda x: alive[min(len(alive) - 1, int(x * len(alive)))]
    if u < 0.08 and alive:                                  # a velocity kick written from outside
        k = np.asarray(sorted({pick(x) for x in r[:1 + int(r[7] * 4)]}), np.uint32)
        st = w.get_body_states(k); st[:, 7:10] += v[:len(k)] * 5.0; w.set_body_states(k, st)
    elif u < 0.12 and len(alive) > 3:                       # an entity leaves (its colliders and joints with it): everything is uploaded again
        e = pick(r[0]); w.destroy_entity(e); alive = [a for a in alive if a != e]
    elif u < 0.18 and alive:                                # forces for one step
        w.apply_force(pick(r[0]), v[0] * 40.0, v[1] * 5.0)
    elif u < 0.20:                                          # state saved and restored in place (mi_world_save_checkpoint / _load_checkpoint)
        w.load_checkpoint(w.save_checkpoint())
    elif u < 0.22 and alive:                                # an entity's joints are removed
        w.destroy_entity_constraints(pick(r[0]))
    elif u < 0.225:
        w.destroy_all_constraints()
    elif u < 0.26:                                          # a new body is dropped in
        e = scenes.make_entities(1); e["position"][0] = (v[0] * 2.0 + np.array([0.0, 6.0 + 3.0 * r[1], 0.0])).astype(np.float32); e["linear_velocity"][0] = v[1] * 3.0
        first = w.create_entities(e)
        c = scenes.make_colliders(1, capi.SPHERE if r[2] < 0.5 else capi.AABB, float(r[3]), float(r[4]), 0.5 + 4.0 * float(r[5]))
        c["shape"][0, :6] = (0, 0, 0, 0.2 + 0.3 * r[6], 0, 0) if r[2] < 0.5 else (-0.3, -0.2 - 0.2 * r[6], -0.25, 0.3, 0.2 + 0.2 * r[6], 0.25)
        w.add_colliders(np.asarray([first], np.uint32), c)
        alive = alive + [int(first)]
    return alive


def run_world(w, sc, steps, plan, events, bodies, record=None):
    """Steps `w`; with record=None returns the recording [(counts, events bytes, states or None)], otherwise compares against it and returns the first difference."""
    s = sc.settings(); out = []
    if events:
        w.enable_events(True)
    alive = list(int(b) for b in bodies)
    for i in range(steps):
        mode, u, r, v = plan[i]
        if mode[0] == "frame":
            w.step(s, mode[1])
        else:
            w.step_fixed(s, sc.dt, 1)
        c = w.counts()
        ev = w.poll_events().tobytes() if events else b""
        st = None
        if (i % 5 == 4 or i == steps - 1) and alive:
            st = w.get_body_states(np.asarray(alive, np.uint32))
        if record is None:
            out.append((c, ev, st))
        else:
            rc, rev, rst = record[i]
            if c != rc:
                return {"step": i, "what": "counts", "gpu": c, "oracle": rc}
            if ev != rev:
                return {"step": i, "what": "events", "gpu_bytes": len(ev), "oracle_bytes": len(rev)}
            if st is not None and st.tobytes() != rst.tobytes():
                bad = np.nonzero((st.view(np.uint32) != rst.view(np.uint32)).any(axis=1))[0]
                return {"step": i, "what": "states", "bodies_differing": int(len(bad)), "first": int(alive[bad[0]]), "max_abs_diff": float(np.nanmax(np.abs(st - rst)))}
        alive = apply_action(w, u, r, v, alive, sc)
    return out if record is None else None


def run_seed(seed, steps, oracle, oracle_only=False, scale=1):
    sc, bodies, rng = make_world_description(seed, scale)
    events = bool(rng.random() < 0.5)
    plan = plan_actions(seed, steps)
    o = sc.populate(oracle.create_world(oracle.ORDER_CANONICAL))
    try:
        record = run_world(o, sc, steps, plan, events, bodies)
    finally:
        o.close()
    # a world that flies apart (random joints can be unstable) is compared up to there only: beyond ~1e6 m the AABBs stop being numbers and "overlap" stops meaning anything
    for i, (c, ev, st) in enumerate(record):
        if st is not None and not (np.abs(st) < 1.0e6).all():
            steps = max(0, (i // 5) * 5 - 5); EXPLODED.append(seed)
            break
    if steps == 0:
        return None
    envs = [{}] + [ENVIRONMENTS[seed % len(ENVIRONMENTS)]]
    for env in envs:
        if PROGRESS:
            with open(PROGRESS, "w") as f:   # (what was running when a device fault took the process down)
                f.write(json.dumps({"seed": seed, "env": env}) + "\n")
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            a = sc.populate(oracle.create_world(oracle.ORDER_CANONICAL) if oracle_only else mi.create_world(0))   # (--oracle-only: the generator's dry run on a machine without a GPU)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        t0 = time.time()
        try:
            r = run_world(a, sc, steps, plan, events, bodies, record)
        finally:
            a.close()
        TIMES.append((round(time.time() - t0, 3), seed, env, steps))
        if r:
            r.update(seed=seed, env=env)
            return r
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:200"); ap.add_argument("--steps", type=int, default=40); ap.add_argument("--budget", type=float, default=600.0)
    ap.add_argument("--out", default=None); ap.add_argument("--oracle-only", action="store_true"); ap.add_argument("--scale", type=int, default=1, help="multiplies the number of entities (6..160)")
    ap.add_argument("--only-terrain", action="store_true", help="skip the worlds whose ground is a box (a fifth of the seeds have a heightmap)")
    args = ap.parse_args()
    import oracle
    oracle.build()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:   # noqa: BLE001
        pass
    global PROGRESS
    PROGRESS = (args.out + ".progress") if args.out else None
    if PROGRESS:
        os.makedirs(os.path.dirname(PROGRESS) or ".", exist_ok=True)
    lo, hi = (int(x) for x in args.seeds.split(":"))
    t0 = time.time(); done = 0; failures = []; errors = []
    for seed in range(lo, hi):
        if time.time() - t0 > args.budget:
            break
        if args.only_terrain and make_world_description(seed, args.scale)[0].heightmap is None:
            continue
        try:
            r = run_seed(seed, args.steps, oracle, args.oracle_only, args.scale)
        except Exception as ex:   # noqa: BLE001 - an API error on either side is a finding too
            errors.append({"seed": seed, "error": repr(ex)[:300], "trace": traceback.format_exc()[-600:]}); r = None
        done += 1
        if r:
            failures.append(r); print("MISMATCH", json.dumps(r), flush=True)
    out = {"seeds": [lo, seed + 1] if args.only_terrain else [lo, lo + done], "only_terrain": args.only_terrain, "steps": args.steps, "worlds": done, "mismatches": failures, "errors": errors, "seconds": round(time.time() - t0, 1)}
    out["exploded_worlds"] = len(EXPLODED); out["slowest"] = sorted(TIMES, key=lambda t: -t[0])[:8]
    print(json.dumps({k: (v if k not in ("mismatches", "errors") else len(v)) for k, v in out.items()}))
    for e in errors[:5]:
        print("ERROR", json.dumps(e)[:900])
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)
    sys.exit(1 if failures or errors else 0)


if __name__ == "__main__":
    main()
