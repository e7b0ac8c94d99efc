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
        for ctype, ea, eb, pod in self.constraints:
            world.add_constraint(ctype, ea, eb, pod)
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


def by_name(name, **kw):
    return {"cfg1": sphere_drop, "cfg2": mixed_stack, "cfg3": obb_pile}[name](**kw)
