"""Seeded synthetic scenes for the BASELINE.json configs.

All randomness comes from a counter-based hash (murmur3 finalizer over (seed, stream, index)), so
a scene is a pure function of (config, size) on every platform and needs no sequential RNG state.
Material constants follow the reference demo scene (src/application.cpp:215,251).
"""
from dataclasses import dataclass, field
import numpy as np
from . import capi


def _hash_u32(seed, stream, idx):
    x = (np.asarray(idx, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed) * np.uint64(0x85EBCA77)
         + np.uint64(stream) * np.uint64(0xC2B2AE3D)) & np.uint64(0xFFFFFFFF)
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x85EBCA6B) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(13)
    x = (x.astype(np.uint64) * np.uint64(0xC2B2AE35) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def uniform(seed, stream, n, lo=0.0, hi=1.0):
    u = _hash_u32(seed, stream, np.arange(n)).astype(np.float64) / 4294967296.0
    return (lo + (hi - lo) * u).astype(np.float32)


def uniform_idx(seed, stream, idx, lo=0.0, hi=1.0):
    """uniform() over explicit counter values (global body indices): tiles of a sharded scene draw identical numbers."""
    u = _hash_u32(seed, stream, np.asarray(idx)).astype(np.float64) / 4294967296.0
    return (lo + (hi - lo) * u).astype(np.float32)


def random_unit_quaternions_idx(seed, stream, idx):
    u1 = uniform_idx(seed, stream, idx).astype(np.float64)
    u2 = uniform_idx(seed, stream + 1, idx).astype(np.float64) * 2 * np.pi
    u3 = uniform_idx(seed, stream + 2, idx).astype(np.float64) * 2 * np.pi
    a, b = np.sqrt(1 - u1), np.sqrt(u1)
    return np.stack([a * np.sin(u2), a * np.cos(u2), b * np.sin(u3), b * np.cos(u3)], axis=1).astype(np.float32)


def random_unit_quaternions(seed, stream, n):
    """Uniform rotations (Shoemake) as x,y,z,w float32."""
    u1 = uniform(seed, stream, n).astype(np.float64)
    u2 = uniform(seed, stream + 1, n).astype(np.float64) * 2 * np.pi
    u3 = uniform(seed, stream + 2, n).astype(np.float64) * 2 * np.pi
    a, b = np.sqrt(1 - u1), np.sqrt(u1)
    q = np.stack([a * np.sin(u2), a * np.cos(u2), b * np.sin(u3), b * np.cos(u3)], axis=1)
    return q.astype(np.float32)


@dataclass
class Scene:
    name: str
    entities: np.ndarray
    collider_entities: np.ndarray
    colliders: np.ndarray
    solver_iterations: int = 30
    dt: float = 1.0 / 120.0
    constraints: list = field(default_factory=list)   # (type, entity_a, entity_b, pod ndarray)
    hulls: list = field(default_factory=list)          # (vertices, triangles)
    global_constraints: list = field(default_factory=list)   # (type, entity_a, entity_b, anchor, axis, limit0, limit1, {field: value})
    forces: list = field(default_factory=list)               # (force-field entity, force3)
    heightmap: dict = None    # chunks_per_dim, chunk_size, restitution, friction, min_corner, amplitude, chunks {(x, z): uint16[129][129]}

    @property
    def num_bodies(self):
        return int((self.entities["kind"] != capi.ENTITY_STATIC).sum())

    def settings(self):
        return capi.StepSettings(1, 120, 4, self.solver_iterations)

    def populate(self, world):
        for v, t in self.hulls:
            world.create_hull_geometry(v, t)
        world.create_entities(self.entities)
        world.add_colliders(self.collider_entities, self.colliders)
        for ent, force in self.forces:
            world.set_force(ent, force)
        if self.heightmap is not None:
            hm = self.heightmap
            world.create_heightmap(hm["chunks_per_dim"], hm["chunk_size"], hm["restitution"], hm["friction"])
            for (x, z), heights in hm["chunks"].items():
                world.set_chunk_heights(x, z, heights)
            world.update_heightmap(hm["min_corner"], hm["amplitude"])
        for ctype, ea, eb, pod in self.constraints:
            world.add_constraint(ctype, ea, eb, pod)
        for ctype, ea, eb, anchor, axis, l0, l1, edits in self.global_constraints:
            cid = world.add_constraint_from_global(ctype, ea, eb, anchor, axis, l0, l1)   # add*ConstraintFromGlobalPoints
            if edits:                                                                      # getConstraint(...).motor... = ...
                pod = world.get_constraint(ctype, cid)
                for k, v in edits.items():
                    pod[k] = v
                world.update_constraint(ctype, cid, pod)
        return world


def make_entities(n, kind=capi.ENTITY_DYNAMIC):
    e = np.zeros(n, dtype=capi.entity_desc)
    e["rotation"][:, 3] = 1.0
    e["gravity_factor"] = 1.0
    e["linear_damping"] = 0.4
    e["angular_damping"] = 0.4
    e["kind"] = kind
    return e


def make_colliders(n, ctype, restitution=0.1, friction=0.5, density=1.0):
    c = np.zeros(n, dtype=capi.collider_desc)
    c["type"] = ctype
    c["restitution"] = restitution
    c["friction"] = friction
    c["density"] = density
    return c


def _lattice(nx, ny, nz, spacing, base_y, seed, jitter):
    # x-major creation order: the reference's first SAP pass (axis 0) then starts nearly sorted.
    ix, iy, iz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    n = nx * ny * nz
    sp = np.broadcast_to(np.asarray(spacing, dtype=np.float32), (3,))
    p = np.stack([(ix.ravel() - (nx - 1) / 2) * sp[0], base_y + iy.ravel() * sp[1], (iz.ravel() - (nz - 1) / 2) * sp[2]], axis=1).astype(np.float32)
    if jitter:
        for a in range(3):
            p[:, a] += uniform(seed, 10 + a, n, -jitter, jitter)
    return p


def _ground(half_xz, thickness=4.0):
    e = make_entities(1, capi.ENTITY_STATIC)
    c = make_colliders(1, capi.AABB, restitution=0.1, friction=1.0, density=4.0)
    c["shape"][0, :6] = (-half_xz, -thickness, -half_xz, half_xz, 0.0, half_xz)
    return e, c


def sphere_drop(n_side=16, seed=1, solver_iterations=10):
    """cfg1: n_side^3 unit spheres (16^3 = 4096) on a jittered lattice over a static 200x4x200 AABB."""
    n = n_side ** 3
    e = make_entities(n)
    e["position"] = _lattice(n_side, n_side, n_side, 2.2, 2.0, seed, 0.05)
    c = make_colliders(n, capi.SPHERE)
    c["shape"][:, 3] = 1.0
    ge, gc = _ground(max(100.0, n_side * 2.2))
    ents = np.concatenate([e, ge])
    cols = np.concatenate([c, gc])
    cent = np.arange(n + 1, dtype=np.uint32)
    return Scene(f"cfg1_sphere_drop_{n}", ents, cent, cols, solver_iterations)


def mixed_stack(nx=64, ny=16, nz=64, seed=2, solver_iterations=30):
    """cfg2: alternating sphere r=0.5 / box h=0.5 (random orientation), lattice spacing 1.1 (+-0.02 jitter)."""
    n = nx * ny * nz
    e = make_entities(n)
    e["position"] = _lattice(nx, ny, nz, 1.1, 0.8, seed, 0.02)
    is_box = (np.arange(n) % 2) == 1
    q = random_unit_quaternions(seed, 20, n)
    e["rotation"][is_box] = q[is_box]
    c = make_colliders(n, capi.SPHERE)
    c["shape"][:, 3] = 0.5
    c["type"][is_box] = capi.AABB
    c["shape"][is_box, :6] = (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5)
    ge, gc = _ground(max(100.0, max(nx, nz) * 1.1))
    return Scene(f"cfg2_mixed_stack_{n}", np.concatenate([e, ge]), np.arange(n + 1, dtype=np.uint32), np.concatenate([c, gc]), solver_iterations)


def obb_pile(nx=128, ny=16, nz=128, seed=3, solver_iterations=20, spacing=1.5):
    """cfg3: nx*ny*nz boxes, half-extents U[0.3,0.6]^3, random orientation, friction 0.5, in a walled pen
    (static ground AABB + 4 static OBB walls)."""
    n = nx * ny * nz
    e = make_entities(n)
    e["position"] = _lattice(nx, ny, nz, spacing, 0.9, seed, 0.02)
    e["rotation"] = random_unit_quaternions(seed, 20, n)
    c = make_colliders(n, capi.AABB, restitution=0.1, friction=0.5)
    h = np.stack([uniform(seed, 30 + a, n, 0.3, 0.6) for a in range(3)], axis=1)
    c["shape"][:, 0:3] = -h
    c["shape"][:, 3:6] = h
    hx, hz = nx * spacing / 2 + 1.0, nz * spacing / 2 + 1.0
    ge, gc = _ground(max(hx, hz) + 10.0)
    we = make_entities(4, capi.ENTITY_STATIC)
    wc = make_colliders(4, capi.OBB, restitution=0.1, friction=0.5)
    wall_h = ny * spacing + 4.0
    centers = [(-hx - 0.5, wall_h / 2, 0), (hx + 0.5, wall_h / 2, 0), (0, wall_h / 2, -hz - 0.5), (0, wall_h / 2, hz + 0.5)]
    radii = [(0.5, wall_h / 2, hz + 1.0), (0.5, wall_h / 2, hz + 1.0), (hx + 1.0, wall_h / 2, 0.5), (hx + 1.0, wall_h / 2, 0.5)]
    for i in range(4):
        wc["shape"][i, 0:4] = (0, 0, 0, 1)
        wc["shape"][i, 4:7] = centers[i]
        wc["shape"][i, 7:10] = radii[i]
    ents = np.concatenate([e, ge, we])
    cols = np.concatenate([c, gc, wc])
    return Scene(f"cfg3_obb_pile_{n}", ents, np.arange(n + 5, dtype=np.uint32), cols, solver_iterations)


def convex_hull_mesh(seed=7, n_points=24, radius=0.6):
    """A small convex hull (vertices, triangles with outward winding) for hull colliders."""
    from scipy.spatial import ConvexHull
    pts = np.stack([uniform(seed, 40 + a, n_points, -1.0, 1.0) for a in range(3)], axis=1).astype(np.float64)
    pts *= radius / np.linalg.norm(pts, axis=1, keepdims=True).clip(0.3)
    hull = ConvexHull(pts)
    verts = pts[hull.vertices].astype(np.float32)
    remap = {v: i for i, v in enumerate(hull.vertices)}
    tris = []
    c = verts.mean(axis=0)
    for simplex in hull.simplices:
        a, b, cc = (remap[v] for v in simplex)
        n = np.cross(verts[b] - verts[a], verts[cc] - verts[a])
        if np.dot(n, verts[a] - c) < 0:
            b, cc = cc, b
        tris.append((a, b, cc))
    return verts, np.asarray(tris, np.uint32)


def shape_zoo(nx=6, ny=4, nz=6, seed=5, solver_iterations=30, spacing=1.6):
    """Every collider type (sphere, capsule, cylinder, AABB->OBB, OBB, hull) in a jittered lattice over the
    ground: exercises all 21 narrow-phase buckets including the GJK/EPA ones."""
    n = nx * ny * nz
    e = make_entities(n)
    e["position"] = _lattice(nx, ny, nz, spacing, 1.0, seed, 0.05)
    e["rotation"] = random_unit_quaternions(seed, 20, n)
    c = make_colliders(n, capi.SPHERE)
    kind = _hash_u32(seed, 50, np.arange(n)) % 6
    r = uniform(seed, 51, n, 0.3, 0.5)
    hl = uniform(seed, 52, n, 0.2, 0.5)
    for i in range(n):
        k = int(kind[i])
        c["type"][i] = k
        if k == capi.SPHERE:
            c["shape"][i, :4] = (0, 0, 0, r[i])
        elif k in (capi.CAPSULE, capi.CYLINDER):
            c["shape"][i, :7] = (0, -hl[i], 0, 0, hl[i], 0, r[i] * 0.7)
        elif k == capi.AABB:
            c["shape"][i, :6] = (-r[i], -hl[i], -r[i] * 0.8, r[i], hl[i], r[i] * 0.8)
        elif k == capi.OBB:
            c["shape"][i, :10] = (0, 0, 0, 1, 0.05, 0, 0, r[i], hl[i], r[i])
        else:
            c["shape"][i, :7] = (0, 0, 0, 1, 0, 0, 0)
            c["hull_geometry"][i] = 0
    ge, gc = _ground(100.0)
    return Scene(f"zoo_{n}", np.concatenate([e, ge]), np.arange(n + 1, dtype=np.uint32), np.concatenate([c, gc]), solver_iterations,
                 hulls=[convex_hull_mesh(seed)])


# ---- quaternion helpers (x, y, z, w) for scene construction -------------------------------------
def q_axis_angle(axis, angle):
    a = np.asarray(axis, np.float64)
    return np.append(a * np.sin(angle * 0.5), np.cos(angle * 0.5))


def q_mul(a, b):
    av, bv = a[:3], b[:3]
    return np.append(av * b[3] + bv * a[3] + np.cross(av, bv), a[3] * b[3] - av.dot(bv))


def q_conj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def q_rot(q, v):
    return q_mul(q_mul(q, np.append(v, 0.0)), q_conj(q))[:3]


def _tangents(n):
    t = np.array([n[1], -n[0], 0.0]) if abs(n[0]) >= 0.57735 else np.array([0.0, n[2], -n[1]])
    t = t / np.linalg.norm(t)
    return t, np.cross(n, t)


_RAGDOLL_SCALE = 0.42
# (name, position, rotation-about-z degrees, colliders) — src/physics/ragdoll.cpp:20-34, 36-105
_RAGDOLL_PARTS = [
    ("torso", (0, 0, 0), 0, [("cap", (-0.2, 0, 0), (0.2, 0, 0), 0.25), ("cap", (-0.16, 0.32, 0), (0.16, 0.32, 0), 0.2),
                             ("cap", (-0.14, 0.62, 0), (0.14, 0.62, 0), 0.22), ("cap", (-0.14, 0.92, 0), (0.14, 0.92, 0), 0.2)]),
    ("head", (0, 1.45, 0), 0, [("cap", (0, -0.075, 0), (0, 0.075, 0), 0.25)]),
    ("l_upper_arm", (-0.6, 0.75, 0), -30, [("cap", (0, -0.2, 0), (0, 0.2, 0), 0.15)]),
    ("l_lower_arm", (-0.884, 0.044, -0.043), -20, [("cap", (0, -0.2, 0), (0, 0.2, 0), 0.15)]),
    ("r_upper_arm", (0.6, 0.75, 0), 30, [("cap", (0, -0.2, 0), (0, 0.2, 0), 0.15)]),
    ("r_lower_arm", (0.884, 0.044, -0.043), 20, [("cap", (0, -0.2, 0), (0, 0.2, 0), 0.15)]),
    ("l_upper_leg", (-0.371, -0.812, 0), -10, [("cap", (0, -0.3, 0), (0, 0.3, 0), 0.25)]),
    ("l_lower_leg", (-0.452, -1.955, 0), -3.5, [("cap", (0, -0.3, 0), (0, 0.3, 0), 0.18)]),
    ("l_foot", (-0.498, -2.585, -0.18), 0, [("box", (0.1587, 0.1, 0.3424))]),
    ("l_toes", (-0.498, -2.585, -0.637), 0, [("cap", (-0.0587, 0, 0), (0.0587, 0, 0), 0.1)]),
    ("r_upper_leg", (0.371, -0.812, 0), 10, [("cap", (0, -0.3, 0), (0, 0.3, 0), 0.25)]),
    ("r_lower_leg", (0.452, -1.955, 0), 3.5, [("cap", (0, -0.3, 0), (0, 0.3, 0), 0.18)]),
    ("r_foot", (0.498, -2.585, -0.18), 0, [("box", (0.1587, 0.1, 0.3424))]),
    ("r_toes", (0.498, -2.585, -0.637), 0, [("cap", (-0.0587, 0, 0), (0.0587, 0, 0), 0.1)]),
]
_P = {name: i for i, (name, *_rest) in enumerate(_RAGDOLL_PARTS)}
_S2 = 1.0 / np.sqrt(2.0)
# (type, a, b, anchor part, local anchor (unscaled), axis spec, limit0 deg, limit1 deg) — ragdoll.cpp:107-123
_RAGDOLL_JOINTS = [
    ("cone", "torso", "head", "torso", (0, 1.2, 0), (0, 1, 0), 50, 90),
    ("cone", "torso", "l_upper_arm", "torso", (-0.4, 1.0, 0), (-1, 0, 0), 130, 90),
    ("hinge", "l_upper_arm", "l_lower_arm", "l_upper_arm", (0, -0.42, 0), (_S2, 0, _S2), -5, 85),
    ("cone", "torso", "r_upper_arm", "torso", (0.4, 1.0, 0), (1, 0, 0), 130, 90),
    ("hinge", "r_upper_arm", "r_lower_arm", "r_upper_arm", (0, -0.42, 0), (_S2, 0, -_S2), -5, 85),
    ("cone", "torso", "l_upper_leg", "torso", (-0.3, -0.25, 0), ("dir", "l_upper_leg", (0, -1, 0)), None, 30),
    ("hinge", "l_upper_leg", "l_lower_leg", "l_upper_leg", (0, -0.6, 0), (1, 0, 0), -90, 5),
    ("cone", "l_lower_leg", "l_foot", "l_lower_leg", (0, -0.52, 0), ("dir", "l_lower_leg", (0, -1, 0)), 75, 20),
    ("hinge", "l_foot", "l_toes", "l_foot", (0, 0, -0.36), (1, 0, 0), -45, 45),
    ("cone", "torso", "r_upper_leg", "torso", (0.3, -0.25, 0), ("dir", "r_upper_leg", (0, -1, 0)), None, 30),
    ("hinge", "r_upper_leg", "r_lower_leg", "r_upper_leg", (0, -0.6, 0), (1, 0, 0), -90, 5),
    ("cone", "r_lower_leg", "r_foot", "r_lower_leg", (0, -0.52, 0), ("dir", "r_lower_leg", (0, -1, 0)), 75, 20),
    ("hinge", "r_foot", "r_toes", "r_foot", (0, 0, -0.36), (1, 0, 0), -45, 45),
]


def _ragdoll_template():
    """Base-pose transforms, colliders and joint PODs of humanoid_ragdoll::initialize (src/physics/ragdoll.cpp:12-123)."""
    sc = _RAGDOLL_SCALE
    pos = [sc * np.asarray(p, np.float64) for _, p, _, _ in _RAGDOLL_PARTS]
    rot = [q_axis_angle((0, 0, 1), np.deg2rad(a)) for _, _, a, _ in _RAGDOLL_PARTS]
    inv_pos = lambda i, g: q_rot(q_conj(rot[i]), g - pos[i])
    inv_dir = lambda i, d: q_rot(q_conj(rot[i]), d)
    joints = []
    for kind, a, b, ap, la, ax, l0, l1 in _RAGDOLL_JOINTS:
        ia, ib, ipn = _P[a], _P[b], _P[ap]
        anchor = q_rot(rot[ipn], sc * np.asarray(la, np.float64)) + pos[ipn]
        axis = q_rot(rot[_P[ax[1]]], np.asarray(ax[2], np.float64)) if isinstance(ax[0], str) else np.asarray(ax, np.float64)
        axA, axB = inv_dir(ia, axis), inv_dir(ib, axis)
        t, bt = _tangents(axA)
        tB = q_rot(q_conj(rot[ib]), q_rot(rot[ia], t))
        if kind == "cone":
            c = np.zeros(1, capi.cone_twist_constraint)
            c["local_anchor_a"], c["local_anchor_b"] = inv_pos(ia, anchor), inv_pos(ib, anchor)
            c["local_limit_axis_a"], c["local_limit_axis_b"] = axA, axB
            c["local_limit_tangent_a"], c["local_limit_bitangent_a"], c["local_limit_tangent_b"] = t, bt, tB
            c["swing_limit"] = -1.0 if l0 is None else np.deg2rad(l0)
            c["twist_limit"] = np.deg2rad(l1)
            c["max_swing_motor_torque"] = -1.0
            c["max_twist_motor_torque"] = -1.0
            joints.append((capi.CONSTRAINT_CONE_TWIST, ia, ib, c))
        else:
            c = np.zeros(1, capi.hinge_constraint)
            c["local_anchor_a"], c["local_anchor_b"] = inv_pos(ia, anchor), inv_pos(ib, anchor)
            c["local_hinge_axis_a"], c["local_hinge_axis_b"] = axA, axB
            c["local_hinge_tangent_a"], c["local_hinge_bitangent_a"], c["local_hinge_tangent_b"] = t, bt, tB
            c["min_rotation_limit"], c["max_rotation_limit"] = np.deg2rad(l0), np.deg2rad(l1)
            c["max_motor_torque"] = -1.0
            joints.append((capi.CONSTRAINT_HINGE, ia, ib, c))
    return pos, rot, joints


def ragdolls(nx=32, nz=32, seed=4, solver_iterations=30, spacing=3.0):
    """cfg4: nx*nz reference ragdolls (14 bodies, 17 colliders, 7 cone-twist + 6 hinge each; density 985, friction 1,
    restitution 0.2) on a grid over the ground, hip height 1.25 + jitter, random yaw."""
    sc = _RAGDOLL_SCALE
    pos, rot, joints = _ragdoll_template()
    nr = nx * nz
    nparts = len(_RAGDOLL_PARTS)
    e = make_entities(nr * nparts)
    ents, cols = [], []
    yaw = uniform(seed, 60, nr, 0.0, 2 * np.pi)
    hip_y = 1.25 + 0.5 * uniform(seed, 61, nr, 0.0, 1.0)
    constraints = []
    for r in range(nr):
        ix, iz = divmod(r, nz)
        hip = np.array([(ix - (nx - 1) / 2) * spacing, hip_y[r], (iz - (nz - 1) / 2) * spacing])
        qy = q_axis_angle((0, 1, 0), yaw[r])
        for i in range(nparts):
            k = r * nparts + i
            e["rotation"][k] = q_mul(qy, rot[i])
            e["position"][k] = q_rot(qy, pos[i]) + hip
            for col in _RAGDOLL_PARTS[i][3]:
                c = make_colliders(1, capi.CAPSULE, restitution=0.2, friction=1.0, density=985.0)
                if col[0] == "cap":
                    c["shape"][0, :7] = (*(sc * np.asarray(col[1])), *(sc * np.asarray(col[2])), sc * col[3])
                else:
                    c["type"] = capi.AABB
                    h = sc * np.asarray(col[1])
                    c["shape"][0, :6] = (*(-h), *h)
                ents.append(k); cols.append(c)
        for ctype, ia, ib, pod in joints:
            constraints.append((ctype, r * nparts + ia, r * nparts + ib, pod))
    ge, gc = _ground(max(100.0, max(nx, nz) * spacing))
    ents.append(nr * nparts); cols.append(gc)
    return Scene(f"cfg4_ragdolls_{nr}", np.concatenate([e, ge]), np.asarray(ents, np.uint32), np.concatenate(cols), solver_iterations,
                 constraints=constraints)


def joint_zoo(seed=6, solver_iterations=30, copies=2):
    """Chains of 3 boxes hanging off kinematic anchors, one chain per joint flavour: distance, ball, fixed, hinge (free /
    limited / velocity motor / position motor), cone-twist (limits / swing+twist motors), slider (limits / motor).
    Covers every init/solve branch of src/physics/constraints.cpp's scalar joint routines."""
    flavours = [
        (capi.CONSTRAINT_DISTANCE, None, 1.0, -1.0, {}),
        (capi.CONSTRAINT_BALL, None, 1.0, -1.0, {}),
        (capi.CONSTRAINT_FIXED, None, 1.0, -1.0, {}),
        (capi.CONSTRAINT_HINGE, (0, 0, 1), 1.0, -1.0, {}),
        (capi.CONSTRAINT_HINGE, (0, 0, 1), -0.3, 0.4, {}),
        (capi.CONSTRAINT_HINGE, (0, 0, 1), 1.0, -1.0, {"max_motor_torque": 40.0, "motor_type": 0, "motor_velocity_or_target_angle": 1.5}),
        (capi.CONSTRAINT_HINGE, (0, 0, 1), -1.0, 1.0, {"max_motor_torque": 60.0, "motor_type": 1, "motor_velocity_or_target_angle": 0.6}),
        (capi.CONSTRAINT_CONE_TWIST, (1, 0, 0), 0.5, 0.3, {}),
        (capi.CONSTRAINT_CONE_TWIST, (1, 0, 0), 0.9, 0.6, {"max_swing_motor_torque": 30.0, "swing_motor_type": 1, "swing_motor_velocity_or_target_angle": 0.4,
                                                         "swing_motor_axis": 0.7, "max_twist_motor_torque": 20.0, "twist_motor_type": 0,
                                                         "twist_motor_velocity_or_target_angle": 1.0}),
        (capi.CONSTRAINT_CONE_TWIST, (1, 0, 0), -1.0, 0.5, {"max_swing_motor_torque": 25.0, "swing_motor_type": 0, "swing_motor_velocity_or_target_angle": 0.8,
                                                          "swing_motor_axis": 0.2, "max_twist_motor_torque": 20.0, "twist_motor_type": 1,
                                                          "twist_motor_velocity_or_target_angle": -0.3}),
        (capi.CONSTRAINT_SLIDER, (1, 0, 0), 1.0, -1.0, {}),
        (capi.CONSTRAINT_SLIDER, (1, 0, 0), -0.2, 0.3, {}),
        (capi.CONSTRAINT_SLIDER, (1, 0, 0), -0.5, 0.5, {"max_motor_force": 200.0, "motor_type": 1, "motor_velocity_or_target_distance": 0.25}),
        (capi.CONSTRAINT_SLIDER, (0, 1, 0), 1.0, -1.0, {"max_motor_force": 150.0, "motor_type": 0, "motor_velocity_or_target_distance": -0.5}),
    ]
    links = 3
    per = 1 + links
    n = len(flavours) * copies * per
    e = make_entities(n)
    c = make_colliders(n, capi.AABB, restitution=0.1, friction=0.6, density=2.0)
    gcs = []
    k = 0
    for cp in range(copies):
        for f, (ctype, axis, l0, l1, edits) in enumerate(flavours):
            base = np.array([f * 2.5 - len(flavours) * 1.25, 5.0 + 0.3 * cp, cp * 6.0 - 3.0], np.float32)
            yaw = q_axis_angle((0, 1, 0), 0.35 * f + 0.9 * cp)
            for l in range(per):
                e["position"][k + l] = base + q_rot(yaw, np.array([0.9 * l, 0, 0]))
                e["rotation"][k + l] = yaw
                c["shape"][k + l, :6] = (-0.35, -0.15, -0.2, 0.35, 0.15, 0.2)
            e["kind"][k] = capi.ENTITY_KINEMATIC
            for l in range(links):
                a, b = k + l, k + l + 1
                anchor = (e["position"][a] + e["position"][b]) * 0.5
                if ctype == capi.CONSTRAINT_DISTANCE:
                    gcs.append((ctype, a, b, e["position"][a].copy(), e["position"][b].copy(), l0, l1, edits))
                else:
                    ax = None if axis is None else q_rot(yaw, np.asarray(axis, np.float64)).astype(np.float32)
                    gcs.append((ctype, a, b, anchor.astype(np.float32), ax, l0, l1, edits))
            k += per
    ge, gc = _ground(100.0)
    return Scene(f"joint_zoo_{n}", np.concatenate([e, ge]), np.arange(n + 1, dtype=np.uint32), np.concatenate([c, gc]), solver_iterations,
                 global_constraints=gcs)


# ---------------------------------------------------------------------------------------------------------------------
# cfg5: vehicles (src/physics/vehicle.cpp:303-497) on convex-hull terrain tiles
# ---------------------------------------------------------------------------------------------------------------------
def _q_from_to(a, b):
    """Shortest-arc rotation a -> b (unit vectors); the rods it orients carry no colliders."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    c = np.cross(a, b); d = float(np.dot(a, b))
    if d < -0.999999:
        ax = np.cross((1.0, 0.0, 0.0), a)
        if np.dot(ax, ax) < 1e-12:
            ax = np.cross((0.0, 1.0, 0.0), a)
        ax = ax / np.linalg.norm(ax)
        return np.array([ax[0], ax[1], ax[2], 0.0])
    q = np.array([c[0], c[1], c[2], 1.0 + d])
    return q / np.linalg.norm(q)


def _gear_teeth(num_teeth, cyl_radius, tooth_length, tooth_width, rod_offset):
    """Capsule colliders of a gear's teeth (vehicle.cpp attach(), attachment_type_gear)."""
    out = []
    for i in range(num_teeth):
        rot = q_axis_angle((0, 1, 0), i * 2.0 * np.pi / num_teeth)
        center = q_rot(rot, np.array([cyl_radius + tooth_length * 0.5, 0.0, 0.0])) + np.array([0.0, rod_offset, 0.0])
        half = q_rot(rot, np.array([tooth_length * 0.5, 0.0, 0.0]))
        out.append(("cap", center - half, center + half, tooth_width * 0.5))
    return out


def _vehicle_template():
    """Bodies, colliders and joints of vehicle::initialize in the vehicle's own frame (motor at the origin)."""
    density = 2000.0
    tl, tw = 0.07, 0.1                      # motorGearDesc tooth length / width
    motor_gear_y, gear_off = 0.25, 0.26
    drive_len, axis_len, susp_len = 4.5, 1.5, 0.4
    wheel_h, wheel_r = 0.3, 0.7
    gear_mat = (0.2, 0.0, density)          # (restitution, friction, density): { wood, 0.2f, desc.friction, desc.density }
    wheel_mat = (0.2, 1.0, 50.0)
    front_z = -drive_len * 0.5 + gear_off * 2.0
    front = np.array([0.0, motor_gear_y + gear_off, front_z])
    steer_rot = q_axis_angle((-1, 0, 0), np.deg2rad(-80.0))
    steer_wheel_pos = np.array([0.0, 1.12, 0.81])
    steer_axis_pos = np.array([0.0, motor_gear_y + gear_off + 0.06, front_z + 0.49])
    steer_axis_len = axis_len * 1.05
    l_att = steer_axis_pos - np.array([steer_axis_len * 0.5, 0, 0]); r_att = steer_axis_pos + np.array([steer_axis_len * 0.5, 0, 0])
    l_susp = front - np.array([axis_len, 0, 0]); r_susp = front + np.array([axis_len, 0, 0])
    l_susp_att = l_susp + np.array([0, 0, susp_len]); r_susp_att = r_susp + np.array([0, 0, susp_len])
    l_fw = l_susp - np.array([susp_len * 0.5, 0, 0]); r_fw = r_susp + np.array([susp_len * 0.5, 0, 0])
    rear_z = drive_len * 0.505
    sun_pos = np.array([-gear_off, motor_gear_y + gear_off, rear_z])
    spider_pos = np.array([0.11, motor_gear_y + gear_off * 2.0, rear_z])
    l_rw = spider_pos + np.array([-gear_off, -gear_off, 0]); r_rw = spider_pos + np.array([gear_off, -gear_off, 0])
    ident = np.array([0.0, 0.0, 0.0, 1.0])
    rot_z90 = q_axis_angle((0, 0, 1), np.deg2rad(90.0)); rot_mz90 = q_axis_angle((0, 0, -1), np.deg2rad(90.0))

    def rod(a, b):
        d = (b - a) / np.linalg.norm(b - a)
        return (a + b) * 0.5, _q_from_to((0, 1, 0), d)

    steer_teeth = []
    stride = (steer_axis_len - tw) / 7.0
    for i in range(8):
        c = np.array([-0.5 * steer_axis_len + 0.5 * tw + i * stride, tw * 0.5, 0.0])
        steer_teeth.append(("cap", c + np.array([0, tl * 0.5, 0]), c - np.array([0, tl * 0.5, 0]), tw * 0.5))
    wheel = lambda off: [("cyl", np.array([0.0, off - wheel_h * 0.5, 0.0]), np.array([0.0, off + wheel_h * 0.5, 0.0]), wheel_r)]
    fa_pos, fa_rot = rod(front + np.array([axis_len, 0, 0]), front - np.array([axis_len, 0, 0]))
    la_pos, la_rot = rod(l_att, l_susp_att); ra_pos, ra_rot = rod(r_att, r_susp_att)
    # (name, position, rotation, colliders, material)
    parts = [
        ("motor", np.zeros(3), ident, [("box", np.array([0.6, 0.1, 1.0]))], gear_mat),
        ("motor_gear", np.array([0.0, motor_gear_y, 0.0]), ident, _gear_teeth(8, 0.2, tl, tw, 0.0), gear_mat),
        ("drive_axis", np.array([0.0, motor_gear_y + gear_off, gear_off]), q_axis_angle((-1, 0, 0), np.deg2rad(90.0)),
         _gear_teeth(8, 0.2, tl, tw, 0.0) + _gear_teeth(8, 0.2, tl, tw, -(drive_len * 0.57 - 1.1)), gear_mat),
        ("front_axis", fa_pos, fa_rot, [], gear_mat),
        ("steering_wheel", steer_wheel_pos, steer_rot, _gear_teeth(8, 0.2, tl, tw, -2.0), gear_mat),
        ("steering_axis", steer_axis_pos, steer_rot, steer_teeth, gear_mat),
        ("l_suspension", l_susp, ident, [], gear_mat),
        ("r_suspension", r_susp, ident, [], gear_mat),
        ("l_front_wheel", l_fw, rot_z90, wheel(0.0), wheel_mat),
        ("r_front_wheel", r_fw, rot_z90, wheel(0.0), wheel_mat),
        ("l_wheel_arm", la_pos, la_rot, [], gear_mat),
        ("r_wheel_arm", ra_pos, ra_rot, [], gear_mat),
        ("sun_gear", sun_pos, rot_mz90, _gear_teeth(17, 0.5, tl, tw, 0.0), gear_mat),
        ("spider_gear", spider_pos, ident, _gear_teeth(8, 0.2, tl, tw, 0.0), gear_mat),
        ("l_rear_wheel", l_rw, rot_mz90, _gear_teeth(8, 0.2, tl, tw, 0.0) + [(k, a, b, r, wheel_mat) for k, a, b, r in wheel(-(axis_len + spider_pos[0]))], gear_mat),
        ("r_rear_wheel", r_rw, rot_mz90, _gear_teeth(8, 0.2, tl, tw, 0.0) + [(k, a, b, r, wheel_mat) for k, a, b, r in wheel(axis_len - spider_pos[0])], gear_mat),
    ]
    P = {name: i for i, (name, *_r) in enumerate(parts)}
    H, B, F, S = capi.CONSTRAINT_HINGE, capi.CONSTRAINT_BALL, capi.CONSTRAINT_FIXED, capi.CONSTRAINT_SLIDER
    d45 = float(np.deg2rad(45.0))
    # (type, a, b, anchor, axis, limit0, limit1, edits)
    joints = [
        (H, "motor", "motor_gear", np.array([0.0, motor_gear_y, 0.0]), (0, 1, 0), 1.0, -1.0, {"max_motor_torque": 500.0, "motor_velocity_or_target_angle": "drive"}),
        (H, "motor", "drive_axis", np.array([0.0, motor_gear_y + gear_off, gear_off]), (0, 0, 1), 1.0, -1.0, {}),
        (F, "motor", "front_axis", front, None, 1.0, -1.0, {}),
        (H, "motor", "steering_wheel", steer_wheel_pos, q_rot(steer_rot, np.array([0.0, -1.0, 0.0])), 1.0, -1.0,
         {"motor_type": 1, "max_motor_torque": 1000.0, "motor_velocity_or_target_angle": 0.0}),
        (S, "motor", "steering_axis", steer_axis_pos, (1, 0, 0), -4.0, 4.0, {}),
        (H, "motor", "l_suspension", l_susp, (0, 1, 0), -d45, d45, {}),
        (H, "motor", "r_suspension", r_susp, (0, 1, 0), -d45, d45, {}),
        (H, "l_front_wheel", "l_suspension", l_fw, (1, 0, 0), 1.0, -1.0, {}),
        (H, "r_front_wheel", "r_suspension", r_fw, (1, 0, 0), 1.0, -1.0, {}),
        (B, "l_suspension", "l_wheel_arm", l_susp_att, None, 1.0, -1.0, {}),
        (B, "steering_axis", "l_wheel_arm", l_att, None, 1.0, -1.0, {}),
        (B, "r_suspension", "r_wheel_arm", r_susp_att, None, 1.0, -1.0, {}),
        (B, "steering_axis", "r_wheel_arm", r_att, None, 1.0, -1.0, {}),
        (H, "motor", "sun_gear", sun_pos, (1, 0, 0), 1.0, -1.0, {}),
        (H, "sun_gear", "spider_gear", spider_pos, (0, 1, 0), 1.0, -1.0, {}),
        (H, "motor", "l_rear_wheel", l_rw, (1, 0, 0), 1.0, -1.0, {}),
        (H, "motor", "r_rear_wheel", r_rw, (1, 0, 0), 1.0, -1.0, {}),
    ]
    return parts, P, joints


def terrain_tile_hull(half=5.0, crown=0.35, thickness=1.0, grid=5):
    """A convex terrain tile: the solid under a concave paraboloid cap over a square base (grid^2 + 4 <= 32 vertices)."""
    from scipy.spatial import ConvexHull
    xs = np.linspace(-half, half, grid)
    top = [(x, -crown * ((x / half) ** 2 + (z / half) ** 2), z) for x in xs for z in xs]
    bottom = [(sx * half, -thickness - 2.0 * crown, sz * half) for sx in (-1, 1) for sz in (-1, 1)]
    pts = np.asarray(top + bottom, np.float64)
    hull = ConvexHull(pts)
    verts = pts[hull.vertices].astype(np.float32)
    remap = {v: i for i, v in enumerate(hull.vertices)}
    c = verts.mean(axis=0)
    tris = []
    for simplex in hull.simplices:
        a, b, cc = (remap[v] for v in simplex)
        n = np.cross(verts[b] - verts[a], verts[cc] - verts[a])
        if np.dot(n, verts[a] - c) < 0:
            b, cc = cc, b
        tris.append((a, b, cc))
    assert len(verts) <= 32
    return verts, np.asarray(tris, np.uint32)


def vehicles(nx=16, nz=16, seed=5, solver_iterations=30, spacing=10.0):
    """cfg5: nx*nz reference vehicles (16 bodies, 86 colliders, 11 hinge + 4 ball + 1 fixed + 1 slider joints each; drive motor
    torque 500, steering position motor 1000) dropped onto static convex-hull terrain tiles (one <= 32-vertex hull geometry,
    one static hull collider per tile, alternating 0 / 90 degree yaw and seeded height offsets)."""
    parts, P, joints = _vehicle_template()
    nv = nx * nz
    npart = len(parts)
    e = make_entities(nv * npart)
    ents, cols, gcs = [], [], []
    yaw = uniform(seed, 70, nv, 0.0, 2 * np.pi)
    drive = uniform(seed, 71, nv, -3.0, 3.0)
    tile_h = uniform(seed, 72, nv, -0.15, 0.15)
    for v in range(nv):
        ix, iz = divmod(v, nz)
        base = np.array([(ix - (nx - 1) / 2) * spacing, tile_h[v] + 1.0, (iz - (nz - 1) / 2) * spacing])
        qy = q_axis_angle((0, 1, 0), yaw[v])
        for i, (name, pos, rot, colliders, mat) in enumerate(parts):
            k = v * npart + i
            e["position"][k] = q_rot(qy, pos) + base
            e["rotation"][k] = q_mul(qy, rot)
            for col in colliders:
                m = col[4] if len(col) > 4 else mat
                c = make_colliders(1, capi.CAPSULE, restitution=m[0], friction=m[1], density=m[2])
                if col[0] == "box":
                    c["type"] = capi.AABB
                    c["shape"][0, :6] = (*(-col[1]), *col[1])
                else:
                    c["type"] = capi.CAPSULE if col[0] == "cap" else capi.CYLINDER
                    c["shape"][0, :7] = (*col[1], *col[2], col[3])
                ents.append(k); cols.append(c)
        for ctype, a, b, anchor, axis, l0, l1, edits in joints:
            ed = {kk: (float(drive[v]) if vv == "drive" else vv) for kk, vv in edits.items()}
            ax = None if axis is None else q_rot(qy, np.asarray(axis, np.float64)).astype(np.float32)
            gcs.append((ctype, v * npart + P[a], v * npart + P[b], (q_rot(qy, anchor) + base).astype(np.float32), ax, l0, l1, ed))
    # terrain
    te = make_entities(nv, capi.ENTITY_STATIC)
    tc = make_colliders(nv, capi.HULL, restitution=0.1, friction=1.0, density=4.0)
    for v in range(nv):
        ix, iz = divmod(v, nz)
        te["position"][v] = ((ix - (nx - 1) / 2) * spacing, tile_h[v], (iz - (nz - 1) / 2) * spacing)
        te["rotation"][v] = q_axis_angle((0, 1, 0), np.pi / 2 * ((ix + iz) & 1))
        tc["shape"][v, :7] = (0, 0, 0, 1, 0, 0, 0)
        tc["hull_geometry"][v] = 0
        ents.append(nv * npart + v)
    cols.append(tc)
    return Scene(f"cfg5_vehicles_{nv}", np.concatenate([e, te]), np.asarray(ents, np.uint32), np.concatenate(cols), solver_iterations,
                 hulls=[terrain_tile_hull(half=spacing / 2)], global_constraints=gcs)


def zones(nx=6, ny=3, nz=6, seed=8, solver_iterations=20, spacing=1.4, localized=True):
    """Triggers and force fields (handleNonCollisionInteractions): a jittered lattice of mixed shapes falls through a wind zone
    (localized force field made of two colliders, tilted entity so the force is rotated), a second overlapping updraft zone, a
    global breeze (force field without colliders) and three trigger volumes of different collider types, onto the ground.
    localized=False leaves the force fields without colliders (all three global)."""
    n = nx * ny * nz
    e = make_entities(n)
    e["position"] = _lattice(nx, ny, nz, spacing, 2.5, seed, 0.05)
    e["rotation"] = random_unit_quaternions(seed, 21, n)
    c = make_colliders(n, capi.SPHERE)
    kind = _hash_u32(seed, 53, np.arange(n)) % 5
    r = uniform(seed, 54, n, 0.25, 0.4)
    for i in range(n):
        k = int(kind[i])
        c["type"][i] = k
        if k == capi.SPHERE: c["shape"][i, :4] = (0, 0, 0, r[i])
        elif k in (capi.CAPSULE, capi.CYLINDER): c["shape"][i, :7] = (0, -r[i], 0, 0, r[i], 0, r[i] * 0.6)
        elif k == capi.AABB: c["shape"][i, :6] = (-r[i], -r[i] * 0.8, -r[i], r[i], r[i] * 0.8, r[i])
        else: c["shape"][i, :10] = (0, 0, 0, 1, 0, 0, 0, r[i], r[i] * 0.7, r[i])
    z = make_entities(6, capi.ENTITY_STATIC)
    z["kind"][:3] = capi.ENTITY_FORCE_FIELD
    z["kind"][3:] = capi.ENTITY_TRIGGER
    z["position"][0] = (-1.5, 1.5, 0.0); z["rotation"][0] = q_axis_angle((0, 0, 1), 0.3)     # wind zone (tilted)
    z["position"][1] = (1.0, 1.0, 1.0)                                                        # updraft
    z["position"][2] = (0.0, 0.0, 0.0)                                                        # global breeze: no colliders
    z["position"][3] = (0.0, 0.8, -1.5); z["position"][4] = (2.0, 0.6, 2.0); z["position"][5] = (-2.0, 1.2, 1.5)
    z["rotation"][5] = q_axis_angle((0, 1, 0), 0.6)
    zc = make_colliders(5, capi.AABB)
    zc["shape"][0, :6] = (-1.5, -1.0, -2.0, 1.5, 1.0, 2.0)
    zc["type"][1] = capi.SPHERE; zc["shape"][1, :4] = (0.5, 1.2, 0.0, 0.9)                      # second collider of the wind zone
    zc["type"][2] = capi.OBB; zc["shape"][2, :10] = (*q_axis_angle((0, 1, 0), 0.4), 0, 0, 0, 1.2, 0.8, 1.2)
    zc["type"][3] = capi.CAPSULE; zc["shape"][3, :7] = (-1.0, 0, 0, 1.0, 0, 0, 0.7)
    zc["shape"][4, :6] = (-1.0, -0.6, -1.0, 1.0, 0.6, 1.0)
    zsphere = make_colliders(1, capi.CYLINDER); zsphere["shape"][0, :7] = (0, -0.8, 0, 0, 0.8, 0, 0.9)
    zent = np.array([n + 0, n + 0, n + 1, n + 3, n + 4, n + 5], np.uint32)
    if not localized:
        zc, zent = zc[3:], zent[3:]
    ge, gc = _ground(100.0)
    ents = np.concatenate([np.arange(n, dtype=np.uint32), zent, [n + 6]]).astype(np.uint32)
    return Scene(f"zones_{n}", np.concatenate([e, z, ge]), ents, np.concatenate([c, zc, zsphere, gc]), solver_iterations,
                 forces=[(n + 0, (6.0, 0.0, 1.0)), (n + 1, (0.0, 14.0, 0.0)), (n + 2, (0.3, 0.0, -0.2))])


def rolling_heightmap(chunks_per_dim=2, chunk_size=16.0, amplitude=6.0, seed=9, min_corner=None, holes=()):
    """Deterministic rolling hills as uint16 chunks (129 x 129 vertices each, shared edge rows).  `holes`: chunks left without
    heights (they collide with nothing, like the reference's unloaded chunks)."""
    n = chunks_per_dim * 128 + 1
    gx, gz = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64))
    ph = uniform(seed, 70, 6).astype(np.float64) * 2 * np.pi
    h = (0.30 + 0.12 * np.sin(gx * 0.045 + ph[0]) * np.cos(gz * 0.038 + ph[1]) + 0.07 * np.sin(gx * 0.11 + gz * 0.09 + ph[2])
         + 0.03 * np.cos(gx * 0.31 + ph[3]) * np.sin(gz * 0.27 + ph[4]))
    q = np.clip(np.rint(h * 65535.0), 0, 65535).astype(np.uint16)
    chunks = {(x, z): np.ascontiguousarray(q[z * 128:z * 128 + 129, x * 128:x * 128 + 129])
              for z in range(chunks_per_dim) for x in range(chunks_per_dim) if (x, z) not in holes}
    half = chunks_per_dim * chunk_size / 2
    corner = (-half, 0.0, -half) if min_corner is None else min_corner
    return dict(chunks_per_dim=chunks_per_dim, chunk_size=chunk_size, restitution=0.05, friction=0.8,
                min_corner=np.asarray(corner, np.float32), amplitude=amplitude, chunks=chunks)


def terrain_field(nx=10, ny=2, nz=10, seed=9, solver_iterations=20, spacing=1.6, with_unsupported=True):
    """Heightmap terrain (heightmapCollision): mixed spheres, capsules, AABB boxes (upright -> AABB, tumbling -> promoted to
    OBB) and OBBs dropped on rolling hills; two of them over the edge of the map (they fall past it); optionally a cylinder
    and a hull (the reference has no triangle routine for them: they are held by their lowest point)."""
    n = nx * ny * nz
    e = make_entities(n)
    e["position"] = _lattice(nx, ny, nz, spacing, 5.5, seed, 0.08)
    rot = random_unit_quaternions(seed, 22, n)
    kind = _hash_u32(seed, 55, np.arange(n)) % 5          # 0 sphere, 1 capsule, 2 upright AABB, 3 tumbling AABB, 4 OBB
    rot[kind == 2] = (0, 0, 0, 1)
    e["rotation"] = rot
    e["angular_velocity"] = (uniform(seed, 57, 3 * n, -1.0, 1.0).reshape(n, 3) * (kind != 2)[:, None]).astype(np.float32)
    e["position"][0] = (-40.0, 8.0, 0.0); e["position"][1] = (3.0, 8.0, 40.0)      # outside the 32 m x 32 m map
    c = make_colliders(n, capi.SPHERE)
    r = uniform(seed, 56, n, 0.25, 0.45)
    for i in range(n):
        k = int(kind[i])
        if k == 0: c["shape"][i, :4] = (0, 0, 0, r[i])
        elif k == 1: c["type"][i] = capi.CAPSULE; c["shape"][i, :7] = (0, -r[i], 0, 0, r[i], 0, r[i] * 0.6)
        elif k in (2, 3): c["type"][i] = capi.AABB; c["shape"][i, :6] = (-r[i], -r[i] * 0.7, -r[i] * 1.2, r[i], r[i] * 0.7, r[i] * 1.2)
        else: c["type"][i] = capi.OBB; c["shape"][i, :10] = (*q_axis_angle((1, 0, 0), 0.5), 0.05, 0, 0, r[i], r[i] * 0.6, r[i] * 0.9)
    ents = np.arange(n, dtype=np.uint32)
    hulls = []
    if with_unsupported:
        x = make_entities(2); x["position"][0] = (0.5, 9.0, 0.5); x["position"][1] = (-1.5, 9.0, 1.0)
        xc = make_colliders(2, capi.CYLINDER); xc["shape"][0, :7] = (0, -0.3, 0, 0, 0.3, 0, 0.3)
        hv, ht = convex_hull_mesh(seed)
        hulls = [(hv, ht)]
        xc["type"][1] = capi.HULL; xc["hull_geometry"][1] = 0; xc["shape"][1, :7] = (0, 0, 0, 1, 0, 0, 0)
        e = np.concatenate([e, x]); c = np.concatenate([c, xc]); ents = np.concatenate([ents, [n, n + 1]]).astype(np.uint32)
    return Scene(f"terrain_field_{n}", e, ents, c, solver_iterations, hulls=hulls, heightmap=rolling_heightmap(seed=seed))


def terrain_wide_colliders(nx=4, nz=4, seed=9, with_unsupported=True):
    """Bodies several times wider than a terrain cell's 64-cell batch on FINE terrain (2 x 2 chunks of 8 m: 6.25 cm cells — a 1 m body spans some 300 cells and the larger
    ones cross chunk borders): the colliders `k_hm_contacts`' large-window instance takes (aligned 8 x 8 blocks in the order of the reference's quadtree walk,
    heightmap_collider.h:35-118), next to small ones the plain instance takes, in one world."""
    sc = terrain_field(nx, 1, nz, seed=seed, spacing=3.4, with_unsupported=with_unsupported)
    sc.heightmap = rolling_heightmap(chunks_per_dim=2, chunk_size=8.0, amplitude=3.0, seed=seed)
    c = sc.colliders
    big = np.arange(len(c)) % 3 != 2                            # every third body keeps its size (a few cells)
    scale = np.where(big, 3.0, 0.35).astype(np.float32)
    for i in range(nx * nz):
        t = int(c["type"][i])
        if t == capi.SPHERE: c["shape"][i, 3] *= scale[i]
        elif t == capi.CAPSULE: c["shape"][i, :7] *= scale[i]
        elif t == capi.AABB: c["shape"][i, :6] *= scale[i]
        else: c["shape"][i, 4:10] *= scale[i]
    sc.entities["position"][0] = (-3.0, 6.0, 0.5); sc.entities["position"][1] = (2.5, 6.0, 3.0)      # (terrain_field parks these two beside ITS map: back over this one)
    sc.name = f"terrain_wide_{nx * nz}"
    return sc


def terrain_big(nx=128, ny=4, nz=128):
    """65 536 mixed bodies on a 4 x 4-chunk (160 m) heightmap: the full-size terrain case."""
    sc = terrain_field(nx, ny, nz, spacing=1.1, with_unsupported=False)
    sc.heightmap = rolling_heightmap(chunks_per_dim=4, chunk_size=40.0, amplitude=8.0)
    return sc


def by_name(name, **kw):
    return {"cfg1": sphere_drop, "cfg2": mixed_stack, "cfg3": obb_pile, "cfg4": ragdolls, "cfg5": vehicles, "zoo": shape_zoo}[name](**kw)


# ---- degenerate / boundary configurations (tests/test_reference_pin.py, tests/test_gpu_parity.py) ----
def scene_from_parts(parts, iterations=30):
    """parts: list of (kind, pos, rot, [(ctype, shape, material overrides)], entity overrides)."""
    ents, cols, cent = [], [], []
    for i, (kind, pos, rot, colliders, over) in enumerate(parts):
        e = make_entities(1, kind); e["position"][0] = pos; e["rotation"][0] = rot
        for k, v in over.items():
            e[k][0] = v
        ents.append(e)
        for ctype, shape, mat in colliders:
            c = make_colliders(1, ctype, **mat); c["shape"][0, :len(shape)] = shape
            cols.append(c); cent.append(i)
    return Scene("edge", np.concatenate(ents), np.asarray(cent, np.uint32), np.concatenate(cols), iterations)


EDGE_CASES = {
    # axis-aligned boxes resting exactly face to face: the SAT's parallel-axes shortcut (absR >= 0.99, collision_narrow.cpp:1219-1226),
    # depths at the slop, AABB (unrotated) next to OBB (a hair rotated: |dot| > 0.99 but not identity)
    "aligned boxes": lambda: scene_from_parts(
        [(capi.ENTITY_DYNAMIC, (0.0, 0.5 + 1.0 * k, 0.0), (0, 0, 0, 1) if k % 2 == 0 else tuple(q_axis_angle((0, 1, 0), 1e-3)),
          [(capi.AABB, (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5), {})], {}) for k in range(6)]
        + [(capi.ENTITY_STATIC, (0, -2.0, 0), (0, 0, 0, 1), [(capi.AABB, (-20, -2, -20, 20, 2, 20), {"friction": 1.0})], {})]),
    # capsules and cylinders lying parallel to each other and to box faces (the |dot| > 0.99 branches of capsule-capsule,
    # capsule-box and cylinder-cylinder, collision_narrow.cpp:532, 623, 830)
    "parallel capsules and cylinders": lambda: scene_from_parts(
        [(capi.ENTITY_DYNAMIC, (0.0, 0.3 + 0.62 * k, 0.0), (0, 0, 0, 1), [(capi.CAPSULE if k % 2 == 0 else capi.CYLINDER, (-0.8, 0, 0, 0.8, 0, 0, 0.3), {})], {}) for k in range(5)]
        + [(capi.ENTITY_DYNAMIC, (3.0, 0.3, 0.0), (0, 0, 0, 1), [(capi.CYLINDER, (0, -0.3, 0, 0, 0.3, 0, 0.5), {})], {}),
           (capi.ENTITY_DYNAMIC, (3.0, 0.95, 0.0), (0, 0, 0, 1), [(capi.CYLINDER, (0, -0.3, 0, 0, 0.3, 0, 0.5), {})], {}),
           (capi.ENTITY_STATIC, (0, -0.5, 0), (0, 0, 0, 1), [(capi.OBB, (0, 0, 0, 1, 0, 0, 0, 20, 0.5, 20), {"friction": 0.9})], {})]),
    # a kinematic platform moving sideways under bodies, a compound body (three colliders of three types), a body without gravity,
    # an undamped spinning body, a body made of one huge and one tiny collider
    "kinematic, compound, odd parameters": lambda: scene_from_parts(
        [(capi.ENTITY_KINEMATIC, (0, 0.0, 0), (0, 0, 0, 1), [(capi.AABB, (-4, -0.25, -4, 4, 0.25, 4), {"friction": 1.0})], {"linear_velocity": (0.8, 0.0, 0.0)}),
         (capi.ENTITY_DYNAMIC, (0.0, 1.2, 0.0), tuple(q_axis_angle((0, 0, 1), 0.3)),
          [(capi.SPHERE, (0.6, 0, 0, 0.35), {}), (capi.CAPSULE, (-0.6, -0.3, 0, -0.6, 0.3, 0, 0.2), {"density": 3.0}), (capi.OBB, (0, 0, 0, 1, 0, 0, 0, 0.5, 0.15, 0.3), {})], {}),
         (capi.ENTITY_DYNAMIC, (2.0, 2.0, 0.5), (0, 0, 0, 1), [(capi.SPHERE, (0, 0, 0, 0.4), {})], {"gravity_factor": 0.0, "linear_velocity": (-1.0, -0.5, 0.0)}),
         (capi.ENTITY_DYNAMIC, (-2.0, 0.8, -1.0), (0, 0, 0, 1), [(capi.AABB, (-0.3, -0.3, -0.3, 0.3, 0.3, 0.3), {"restitution": 0.9})],
          {"linear_damping": 0.0, "angular_damping": 0.0, "angular_velocity": (0.0, 9.0, 3.0)}),
         (capi.ENTITY_DYNAMIC, (1.0, 1.5, -2.0), (0, 0, 0, 1), [(capi.AABB, (-1.0, -0.2, -1.0, 1.0, 0.2, 1.0), {}), (capi.SPHERE, (0, 0.25, 0, 0.05), {"density": 50.0})], {}),
         (capi.ENTITY_STATIC, (0, -3.0, 0), (0, 0, 0, 1), [(capi.AABB, (-30, -1, -30, 30, 1, 30), {})], {})]),
    # nothing dynamic touches anything; then only static colliders; both must step without contacts
    "free flight only": lambda: scene_from_parts(
        [(capi.ENTITY_DYNAMIC, (3.0 * k, 50.0, 0), tuple(q_axis_angle((1, 0, 0), 0.2 * k)), [(k % 6 if k % 6 != capi.HULL else capi.SPHERE,
          {0: (0, 0, 0, 0.5), 1: (0, -0.4, 0, 0, 0.4, 0, 0.2), 2: (0, -0.4, 0, 0, 0.4, 0, 0.3), 3: (-0.3, -0.4, -0.5, 0.3, 0.4, 0.5), 4: (0, 0, 0, 1, 0, 0, 0, 0.3, 0.4, 0.5), 5: (0, 0, 0, 0.5)}[k % 6], {})],
          {"angular_velocity": (0.5 * k, 1.0, -0.3 * k)}) for k in range(7)]),
}
