"""ctypes binding of the C ABI declared in include/mi_physics.h.

The same binding class serves any shared library exporting that ABI under a symbol prefix
(`mi_` for the HIP product library; the test oracle re-uses it with its own prefix from
oracle/__init__.py — the product never imports the oracle).
"""
import ctypes as C
import numpy as np

MI_OK = 0
ENTITY_DYNAMIC, ENTITY_KINEMATIC, ENTITY_STATIC, ENTITY_TRIGGER, ENTITY_FORCE_FIELD = 0, 1, 2, 3, 4
SPHERE, CAPSULE, CYLINDER, AABB, OBB, HULL = range(6)
(CONSTRAINT_DISTANCE, CONSTRAINT_BALL, CONSTRAINT_FIXED, CONSTRAINT_HINGE,
 CONSTRAINT_CONE_TWIST, CONSTRAINT_SLIDER) = range(6)

# numpy mirrors of the POD structs (all 4-byte members, no padding).
entity_desc = np.dtype([
    ("position", "<f4", 3), ("rotation", "<f4", 4),
    ("linear_velocity", "<f4", 3), ("angular_velocity", "<f4", 3),
    ("gravity_factor", "<f4"), ("linear_damping", "<f4"), ("angular_damping", "<f4"),
    ("kind", "<u4")])
collider_desc = np.dtype([
    ("type", "<u4"), ("object_type", "<u4"), ("shape", "<f4", 12), ("hull_geometry", "<u4"),
    ("restitution", "<f4"), ("friction", "<f4"), ("density", "<f4")])
EVENT_COLLISION_BEGIN, EVENT_COLLISION_END, EVENT_TRIGGER_ENTER, EVENT_TRIGGER_LEAVE = 0, 1, 2, 3
event_dtype = np.dtype([("type", "<u4"), ("entity_a", "<u4"), ("entity_b", "<u4"), ("collider_a", "<u4"), ("collider_b", "<u4"),
                        ("point", "<f4", 3), ("normal", "<f4", 3), ("relative_velocity", "<f4", 3)])
contact_dtype = np.dtype([
    ("point", "<f4", 3), ("penetration_depth", "<f4"), ("normal", "<f4", 3),
    ("friction_restitution", "<u4"), ("collider_a", "<u4"), ("collider_b", "<u4"),
    ("body_a", "<u4"), ("body_b", "<u4")])
assert entity_desc.itemsize == 68 and collider_desc.itemsize == 72 and contact_dtype.itemsize == 48

distance_constraint = np.dtype([("local_anchor_a", "<f4", 3), ("local_anchor_b", "<f4", 3), ("global_length", "<f4")])
ball_constraint = np.dtype([("local_anchor_a", "<f4", 3), ("local_anchor_b", "<f4", 3)])
cloth_desc = np.dtype([("width", "<f4"), ("height", "<f4"), ("grid_size_x", "<u4"), ("grid_size_y", "<u4"), ("total_mass", "<f4"), ("stiffness", "<f4"),
                       ("damping", "<f4"), ("gravity_factor", "<f4")])
fixed_constraint = np.dtype([("initial_inv_rotation_difference", "<f4", 4), ("local_anchor_a", "<f4", 3), ("local_anchor_b", "<f4", 3)])
hinge_constraint = np.dtype([
    ("local_anchor_a", "<f4", 3), ("local_anchor_b", "<f4", 3), ("local_hinge_axis_a", "<f4", 3), ("local_hinge_axis_b", "<f4", 3),
    ("min_rotation_limit", "<f4"), ("max_rotation_limit", "<f4"), ("max_motor_torque", "<f4"), ("motor_type", "<u4"),
    ("motor_velocity_or_target_angle", "<f4"),
    ("local_hinge_tangent_a", "<f4", 3), ("local_hinge_bitangent_a", "<f4", 3), ("local_hinge_tangent_b", "<f4", 3)])
cone_twist_constraint = np.dtype([
    ("local_anchor_a", "<f4", 3), ("local_anchor_b", "<f4", 3), ("local_limit_axis_a", "<f4", 3), ("local_limit_axis_b", "<f4", 3),
    ("local_limit_tangent_a", "<f4", 3), ("local_limit_bitangent_a", "<f4", 3), ("local_limit_tangent_b", "<f4", 3),
    ("swing_limit", "<f4"), ("twist_limit", "<f4"),
    ("swing_motor_type", "<u4"), ("swing_motor_velocity_or_target_angle", "<f4"), ("max_swing_motor_torque", "<f4"), ("swing_motor_axis", "<f4"),
    ("twist_motor_type", "<u4"), ("twist_motor_velocity_or_target_angle", "<f4"), ("max_twist_motor_torque", "<f4")])
slider_constraint = np.dtype([
    ("initial_inv_rotation_difference", "<f4", 4), ("local_anchor_a", "<f4", 3), ("local_anchor_b", "<f4", 3), ("local_axis_a", "<f4", 3),
    ("neg_distance_limit", "<f4"), ("pos_distance_limit", "<f4"), ("max_motor_force", "<f4"), ("motor_type", "<u4"),
    ("motor_velocity_or_target_distance", "<f4")])
CONSTRAINT_DTYPES = [distance_constraint, ball_constraint, fixed_constraint, hinge_constraint, cone_twist_constraint, slider_constraint]


class StepSettings(C.Structure):
    """physics_settings minus callbacks (src/physics/physics.h:382-400)."""
    _fields_ = [("fixed_frame_rate", C.c_uint32), ("frame_rate", C.c_uint32),
                ("max_physics_iterations_per_frame", C.c_uint32), ("num_rigid_solver_iterations", C.c_uint32)]

    def __init__(self, fixed_frame_rate=1, frame_rate=120, max_physics_iterations_per_frame=4, num_rigid_solver_iterations=30):
        super().__init__(fixed_frame_rate, frame_rate, max_physics_iterations_per_frame, num_rigid_solver_iterations)


class StepCounts(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_rigid_bodies", "num_colliders", "num_broadphase_overlaps", "num_collisions",
                                          "num_contacts", "num_colors", "sorting_axis", "reserved")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}

    @property
    def solve_launches(self):
        return self.reserved


class StageTimes(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("world_colliders", "broadphase", "narrowphase", "integrate_forces", "schedule",
                                         "init_constraints", "solve", "integrate_velocities", "total")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class ShardDesc(C.Structure):
    """mi_shard_desc (include/mi_shard.h)."""
    _fields_ = [("rank", C.c_uint32), ("num_ranks", C.c_uint32), ("origin_x", C.c_float), ("origin_z", C.c_float), ("tile_size_x", C.c_float),
                ("tile_size_z", C.c_float), ("tiles_x", C.c_uint32), ("tiles_z", C.c_uint32), ("ghost_margin", C.c_float), ("max_records", C.c_uint32)]


SWEEP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32)   # mi_shard_sweep_fn


class ShardExchangeStats(C.Structure):
    """mi_shard_exchange_stats (include/mi_shard.h)."""
    _fields_ = [("exchanges", C.c_uint64), ("device_ms_sum", C.c_double), ("message_bytes", C.c_uint64), ("num_neighbours", C.c_uint32), ("library_transport", C.c_uint32),
                ("neighbour_rank", C.c_uint32 * 8), ("records_last", C.c_uint32 * 8), ("records_sum", C.c_uint64 * 8), ("owned_bodies", C.c_uint32), ("ghost_bodies", C.c_uint32),
                ("sweep_exchanges", C.c_uint64), ("sweep_message_bytes", C.c_uint64), ("sweep_records_last", C.c_uint32 * 8),
                ("message_records_last", C.c_uint32 * 8), ("message_bytes_sum", C.c_uint64)]


SHARD_RECORD_FLOATS = 14


class WorldDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32)]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class PhysicsError(RuntimeError):
    pass


class Library:
    """A loaded shared library exporting the ABI under `prefix`."""

    def __init__(self, path, prefix="mi_"):
        self.path = str(path)
        self.prefix = prefix
        self.lib = C.CDLL(self.path)

    def fn(self, name, restype=C.c_int):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        return f

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)

    def last_error(self):
        if self.has("last_error"):
            s = self.fn("last_error", C.c_char_p)()
            return s.decode() if s else ""
        return ""

    def shard_unique_id(self):
        """ncclGetUniqueId through the library (rank 0; distribute the 128 bytes to every rank)."""
        buf = np.zeros(128, np.uint8)
        self.check(self.fn("shard_get_unique_id")(_ptr(buf)), "shard_get_unique_id")
        return buf.tobytes()

    def shard_library_transport_available(self):
        """Non-collective probe: can this process use the library's RCCL transport?  (Agree on it over all ranks before anyone attaches.)"""
        return bool(self.fn("shard_library_transport_available")()) if self.has("shard_library_transport_available") else False

    def shard_tile_of_rank(self, tiles_x, tiles_z, rank):
        out = C.c_uint32()
        self.check(self.fn("shard_tile_of_rank")(C.c_uint32(tiles_x), C.c_uint32(tiles_z), C.c_uint32(rank), C.byref(out)), "shard_tile_of_rank")
        return out.value

    def shard_balance_borders(self, hist, lo, hi, tiles, cur, margin):
        """Borders (tiles - 1) that even out the body counts of `hist` (bins over [lo, hi), summed over all ranks), clamped to what one
        change may do from `cur` (include/mi_shard.h); the same on every rank for the same input."""
        h = np.ascontiguousarray(hist, np.uint64); c = np.ascontiguousarray(cur, np.float32); out = np.zeros(max(tiles - 1, 1), np.float32)
        self.check(self.fn("shard_balance_borders")(_ptr(h), C.c_uint32(len(h)), C.c_float(lo), C.c_float(hi), C.c_uint32(tiles), _ptr(c) if tiles > 1 else None,
                                                    C.c_float(margin), _ptr(out) if tiles > 1 else None), "shard_balance_borders")
        return out[: tiles - 1]

    def check(self, rc, what):
        if rc != MI_OK:
            raise PhysicsError(f"{self.prefix}{what} failed with status {rc}: {self.last_error()}")


class World:
    """Thin object wrapper over one `mi_world*` (or oracle world) handle."""

    def __init__(self, library, handle):
        self.L = library
        self.h = handle
        self._n_entities = 0

    def close(self):
        if self.h:
            self.L.fn("world_destroy", None)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- scene construction
    def create_entities(self, descs):
        descs = np.ascontiguousarray(descs, dtype=entity_desc)
        first = C.c_uint32(0)
        self.L.check(self.L.fn("entities_create")(self.h, C.c_uint32(len(descs)), _ptr(descs), C.byref(first)), "entities_create")
        self._n_entities += len(descs)
        return first.value

    def destroy_entity(self, entity):
        """game_scene::deleteEntity(entity) — src/scene/scene.cpp:124-150."""
        self.L.check(self.L.fn("entity_destroy")(self.h, C.c_uint32(entity)), "entity_destroy")

    def add_colliders(self, entities, descs):
        entities = np.ascontiguousarray(entities, dtype=np.uint32)
        descs = np.ascontiguousarray(descs, dtype=collider_desc)
        assert len(entities) == len(descs)
        self.L.check(self.L.fn("colliders_add")(self.h, C.c_uint32(len(descs)), _ptr(entities), _ptr(descs)), "colliders_add")

    def create_hull_geometry(self, vertices, triangles):
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        out = C.c_uint32(0)
        self.L.check(self.L.fn("hull_geometry_create")(self.h, _ptr(v), C.c_uint32(len(v)), _ptr(t), C.c_uint32(len(t)), C.byref(out)),
                     "hull_geometry_create")
        return out.value

    def add_constraint(self, ctype, entity_a, entity_b, pod):
        pod = np.ascontiguousarray(pod, dtype=CONSTRAINT_DTYPES[ctype]).reshape(1)
        out = C.c_uint32(0)
        self.L.check(self.L.fn("constraint_create")(self.h, C.c_uint32(ctype), C.c_uint32(entity_a), C.c_uint32(entity_b),
                                                    _ptr(pod), C.c_uint32(pod.dtype.itemsize), C.byref(out)), "constraint_create")
        return out.value

    def add_constraint_from_global(self, ctype, entity_a, entity_b, anchor, axis=None, limit0=1.0, limit1=-1.0):
        a = np.ascontiguousarray(anchor, dtype=np.float32)
        x = np.ascontiguousarray(axis if axis is not None else (0, 0, 0), dtype=np.float32)
        out = C.c_uint32(0)
        self.L.check(self.L.fn("constraint_create_from_global")(self.h, C.c_uint32(ctype), C.c_uint32(entity_a), C.c_uint32(entity_b),
                                                                _ptr(a), _ptr(x), C.c_float(limit0), C.c_float(limit1), C.byref(out)),
                     "constraint_create_from_global")
        return out.value

    def get_constraint(self, ctype, cid):
        pod = np.zeros(1, dtype=CONSTRAINT_DTYPES[ctype])
        self.L.check(self.L.fn("constraint_get")(self.h, C.c_uint32(ctype), C.c_uint32(cid), _ptr(pod), C.c_uint32(pod.dtype.itemsize)),
                     "constraint_get")
        return pod

    def destroy_constraint(self, ctype, cid):
        """deleteConstraint(scene, handle) — src/physics/physics.cpp:474-521."""
        self.L.check(self.L.fn("constraint_destroy")(self.h, C.c_uint32(ctype), C.c_uint32(cid)), "constraint_destroy")

    def destroy_all_constraints(self):
        self.L.check(self.L.fn("constraints_destroy_all")(self.h), "constraints_destroy_all")

    def destroy_entity_constraints(self, entity):
        """deleteAllConstraintsFromEntity(entity) — src/physics/physics.cpp:523-539."""
        self.L.check(self.L.fn("entity_destroy_constraints")(self.h, C.c_uint32(entity)), "entity_destroy_constraints")

    def update_constraint(self, ctype, cid, pod):
        pod = np.ascontiguousarray(pod, dtype=CONSTRAINT_DTYPES[ctype]).reshape(1)
        self.L.check(self.L.fn("constraint_update")(self.h, C.c_uint32(ctype), C.c_uint32(cid), _ptr(pod), C.c_uint32(pod.dtype.itemsize)),
                     "constraint_update")

    def apply_force(self, entity, force=None, torque=None):
        f = np.ascontiguousarray(force, dtype=np.float32) if force is not None else None
        t = np.ascontiguousarray(torque, dtype=np.float32) if torque is not None else None
        self.L.check(self.L.fn("entity_apply_force")(self.h, C.c_uint32(entity), _ptr(f), _ptr(t)), "entity_apply_force")

    def apply_forces(self, entities, forces=None, torques=None):
        """rb.forceAccumulator += f; rb.torqueAccumulator += tau for many bodies, in order."""
        e = np.ascontiguousarray(entities, dtype=np.uint32)
        f = np.ascontiguousarray(forces, dtype=np.float32).reshape(len(e), 3) if forces is not None else None
        t = np.ascontiguousarray(torques, dtype=np.float32).reshape(len(e), 3) if torques is not None else None
        self.L.check(self.L.fn("entities_apply_forces")(self.h, C.c_uint32(len(e)), _ptr(e), _ptr(f), _ptr(t)), "entities_apply_forces")

    def test_interactions(self, origins, directions, strengths=None, entity_ranges=None):
        """testPhysicsInteraction(scene, ray, strength) for several rays (src/physics/physics.cpp:555-629)."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        s = np.ascontiguousarray(strengths, dtype=np.float32) if strengths is not None else None
        r = np.ascontiguousarray(entity_ranges, dtype=np.uint32).reshape(-1, 2) if entity_ranges is not None else None
        self.L.check(self.L.fn("world_test_interactions")(self.h, C.c_uint32(len(o)), _ptr(o), _ptr(d), _ptr(s), _ptr(r)), "world_test_interactions")

    def update_constraints(self, ctype, ids, pods):
        """getConstraint(scene, handle) = ... for many constraints of one type."""
        i = np.ascontiguousarray(ids, dtype=np.uint32)
        p = np.ascontiguousarray(pods)
        assert len(p) == len(i)
        self.L.check(self.L.fn("constraints_update")(self.h, C.c_uint32(ctype), C.c_uint32(len(i)), _ptr(i), _ptr(p), C.c_uint32(p.dtype.itemsize)), "constraints_update")

    def set_force(self, entity, force):
        """force_field_component::force of a FORCE_FIELD entity (entity-local frame)."""
        f = np.ascontiguousarray(force, dtype=np.float32)
        self.L.check(self.L.fn("entity_set_force")(self.h, C.c_uint32(entity), _ptr(f)), "entity_set_force")

    # --- heightmap terrain (heightmap_collider_component, src/terrain/heightmap_collider.h:126-151)
    def create_heightmap(self, chunks_per_dim, chunk_size, restitution=0.0, friction=1.0):
        self.L.check(self.L.fn("heightmap_create")(self.h, C.c_uint32(chunks_per_dim), C.c_float(chunk_size), C.c_float(restitution), C.c_float(friction)), "heightmap_create")

    def set_chunk_heights(self, x, z, heights):
        """collider(x, z).setHeights: 129 x 129 uint16, [z][x]."""
        h = np.ascontiguousarray(heights, dtype=np.uint16)
        assert h.shape == (129, 129)
        self.L.check(self.L.fn("heightmap_set_chunk_heights")(self.h, C.c_uint32(x), C.c_uint32(z), _ptr(h)), "heightmap_set_chunk_heights")

    def update_heightmap(self, min_corner, amplitude_scale):
        c = np.ascontiguousarray(min_corner, dtype=np.float32)
        self.L.check(self.L.fn("heightmap_update")(self.h, _ptr(c), C.c_float(amplitude_scale)), "heightmap_update")

    def heightmap_height(self, x, z):
        out = C.c_float()
        self.L.check(self.L.fn("heightmap_get_height")(self.h, C.c_float(x), C.c_float(z), C.byref(out)), "heightmap_get_height")
        return out.value

    # --- cloth (cloth_component, src/physics/cloth.h:5-60)
    def create_cloth(self, width, height, grid_x, grid_y, total_mass, stiffness=0.5, damping=0.3, gravity_factor=1.0):
        d = np.zeros(1, cloth_desc)
        d["width"], d["height"], d["grid_size_x"], d["grid_size_y"] = width, height, grid_x, grid_y
        d["total_mass"], d["stiffness"], d["damping"], d["gravity_factor"] = total_mass, stiffness, damping, gravity_factor
        out = C.c_uint32()
        self.L.check(self.L.fn("cloth_create")(self.h, _ptr(d), C.byref(out)), "cloth_create")
        return out.value

    def set_cloth_fixed_vertices(self, cloth, position, rotation=(0, 0, 0, 1), move_rigid=False):
        p = np.ascontiguousarray(position, np.float32); r = np.ascontiguousarray(rotation, np.float32)
        self.L.check(self.L.fn("cloth_set_fixed_vertices")(self.h, C.c_uint32(cloth), _ptr(p), _ptr(r), C.c_uint32(1 if move_rigid else 0)), "cloth_set_fixed_vertices")

    def set_cloth_properties(self, cloth, total_mass, stiffness, damping, gravity_factor):
        self.L.check(self.L.fn("cloth_set_properties")(self.h, C.c_uint32(cloth), C.c_float(total_mass), C.c_float(stiffness), C.c_float(damping), C.c_float(gravity_factor)),
                     "cloth_set_properties")

    def cloth_state(self, cloth, num_particles):
        pos = np.zeros((num_particles, 3), np.float32); vel = np.zeros((num_particles, 3), np.float32)
        self.L.check(self.L.fn("cloth_get_state")(self.h, C.c_uint32(cloth), _ptr(pos), _ptr(vel), C.c_uint32(num_particles)), "cloth_get_state")
        return pos, vel

    def set_cloth_iterations(self, velocity=0, position=1, drift=0):
        self.L.check(self.L.fn("world_set_cloth_iterations")(self.h, C.c_uint32(velocity), C.c_uint32(position), C.c_uint32(drift)), "world_set_cloth_iterations")

    def serialize_entity_native(self, entity):
        """serializeEntityToMemory written by the library itself (only the reference build under oracle/_ref exports it: its stream is
        made from the reference's own struct definitions and pins d3d12renderer_amd/scene_binary.py)."""
        size = C.c_uint64()
        buf = np.zeros(1 << 16, np.uint8)
        self.L.check(self.L.fn("entity_serialize")(self.h, C.c_uint32(entity), _ptr(buf), C.c_uint64(len(buf)), C.byref(size)), "entity_serialize")
        return buf[: size.value].tobytes()

    # --- checkpoint / resume
    def debug_set_sweep_axis(self, axis):
        self.L.check(self.L.fn("debug_set_sweep_axis")(self.h, C.c_uint32(axis)), "debug_set_sweep_axis")

    def debug_set_solve_order(self, pairs):
        """The next internal step solves exactly these oriented collider pairs ((n, 2) world indices), sequentially, in this order (and the
        joints in pool order): the reference's own constraint order (include/mi_physics.h)."""
        p = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        self.L.check(self.L.fn("debug_set_solve_order")(self.h, _ptr(p) if len(p) else None, C.c_uint32(len(p))), "debug_set_solve_order")

    def debug_step_ahead_stats(self):
        """mi_debug_step_ahead_stats: (times the next step's first kernel was enqueued ahead, times the next step adopted it)."""
        a, b = C.c_uint64(), C.c_uint64()
        self.L.check(self.L.fn("debug_step_ahead_stats")(self.h, C.byref(a), C.byref(b)), "debug_step_ahead_stats")
        return a.value, b.value

    def debug_set_solve_dataflow(self, enable=True):
        """mi_debug_set_solve_dataflow: steps that follow a caller's order run it through the production contact solver (levels of the order as colours)."""
        self.L.check(self.L.fn("debug_set_solve_dataflow")(self.h, C.c_uint32(1 if enable else 0)), "debug_set_solve_dataflow")

    def debug_solve_order_depth(self):
        d = C.c_uint32()
        self.L.check(self.L.fn("debug_solve_order_depth")(self.h, C.byref(d)), "debug_solve_order_depth")
        return d.value

    def save_checkpoint(self):
        size = C.c_uint64()
        self.L.check(self.L.fn("world_save_checkpoint")(self.h, None, C.c_uint64(0), C.byref(size)), "world_save_checkpoint")
        buf = np.zeros(size.value, np.uint8)
        self.L.check(self.L.fn("world_save_checkpoint")(self.h, _ptr(buf), C.c_uint64(len(buf)), C.byref(size)), "world_save_checkpoint")
        return buf.tobytes()

    def load_checkpoint(self, blob):
        buf = np.frombuffer(blob, np.uint8)
        self.L.check(self.L.fn("world_load_checkpoint")(self.h, _ptr(buf), C.c_uint64(len(buf))), "world_load_checkpoint")

    # --- stepping
    def step(self, settings, dt):
        """physicsStep(scene, arena, timer, settings, dt) — src/physics/physics.cpp:1364."""
        self.L.check(self.L.fn("world_step")(self.h, C.byref(settings), C.c_float(dt)), "world_step")

    def step_fixed(self, settings, dt, n=1):
        """n x physicsStepInternal — src/physics/physics.cpp:1180."""
        # the hot call of every stepping loop: bound once, arguments converted once per (settings, dt, n) — the generic path costs ~6 us of Python per call, which
        # a loop that steps a 1 ms world one step per call pays as idle GPU between two steps
        key = (id(settings), dt, n)
        if getattr(self, "_step_key", None) != key:
            f = self.L.fn("world_step_fixed")
            self._step_call = (f, self.h, C.byref(settings), C.c_float(dt), C.c_uint32(n), settings)   # (settings kept alive)
            self._step_key = key
        f, h, ps, cdt, cn, _ = self._step_call
        rc = f(h, ps, cdt, cn)
        if rc != MI_OK:
            self.L.check(rc, "world_step_fixed")

    def step_profiled(self, settings, dt):
        """One internal step with per-launch HIP events around the dominant kernel -> (launches, kernel_ms, contact_updates)."""
        n = C.c_uint32(0); ms = C.c_float(0); upd = C.c_uint64(0)
        self.L.check(self.L.fn("world_step_profiled")(self.h, C.byref(settings), C.c_float(dt), C.byref(n), C.byref(ms), C.byref(upd)), "world_step_profiled")
        return n.value, ms.value, upd.value

    def enable_events(self, enable=True):
        self.L.check(self.L.fn("world_enable_events")(self.h, C.c_uint32(1 if enable else 0)), "world_enable_events")

    def poll_events(self):
        """Collision begin / end events of the internal steps since the last poll (numpy structured array)."""
        n = C.c_uint32(0)
        self.L.check(self.L.fn("world_poll_events")(self.h, None, C.c_uint32(0), C.byref(n)), "world_poll_events")
        out = np.zeros(n.value, dtype=event_dtype)
        if n.value:
            self.L.check(self.L.fn("world_poll_events")(self.h, _ptr(out), C.c_uint32(n.value), C.byref(n)), "world_poll_events")
        return out

    def color_tail_stats(self):
        """(valid steps whose colouring was finished inside k_bin_hist, colouring rounds run there) since creation (product library only)."""
        a = C.c_uint64(0); b = C.c_uint64(0)
        self.L.check(self.L.fn("debug_color_tail_stats")(self.h, C.byref(a), C.byref(b)), "debug_color_tail_stats")
        return a.value, b.value

    def step_mode_stats(self):
        """(internal steps, speculative steps, synchronous retries) since creation (product library only)."""
        a = C.c_uint32(0); b = C.c_uint32(0); c = C.c_uint32(0)
        self.L.check(self.L.fn("world_get_step_mode_stats")(self.h, C.byref(a), C.byref(b), C.byref(c)), "world_get_step_mode_stats")
        return a.value, b.value, c.value

    # --- read-back
    def num_entities(self):
        n = C.c_uint32(0)
        self.L.check(self.L.fn("world_num_entities")(self.h, C.byref(n)), "world_num_entities")
        return n.value

    def _get2(self, name, wa, wb):
        n = self.num_entities()
        a = np.zeros((n, wa), np.float32)
        b = np.zeros((n, wb), np.float32)
        self.L.check(self.L.fn(name)(self.h, _ptr(a), _ptr(b), C.c_uint32(n)), name)
        return a, b

    def transforms(self):
        return self._get2("world_get_transforms", 3, 4)

    def physics_transforms(self):
        return self._get2("world_get_physics_transforms", 3, 4)

    def transforms_view(self, physics=False):
        """mi_world_view_transforms / mi_world_view_physics_transforms: (positions [n, 3], rotations [n, 4]) as read-only numpy views of the
        library's pinned host rows — no copy on the host side; valid until the second next stepping call (product library only)."""
        pp = C.POINTER(C.c_float)(); rr = C.POINTER(C.c_float)(); n = C.c_uint32()
        name = "world_view_physics_transforms" if physics else "world_view_transforms"
        self.L.check(self.L.fn(name)(self.h, C.byref(pp), C.byref(rr), C.byref(n)), name)
        p = np.ctypeslib.as_array(pp, shape=(n.value, 3)); r = np.ctypeslib.as_array(rr, shape=(n.value, 4))
        p.flags.writeable = False; r.flags.writeable = False
        return p, r

    def transforms_view_landed(self, physics=False):
        """mi_world_view_transforms_landed: (positions, rotations, internal step they belong to) — the newest pose rows that are complete in host memory, without waiting for
        a copy still on the bus (one frame behind in a step -> view loop); read-only views, valid until the next stepping call (product library only)."""
        pp = C.POINTER(C.c_float)(); rr = C.POINTER(C.c_float)(); n = C.c_uint32(); st = C.c_uint64()
        self.L.check(self.L.fn("world_view_transforms_landed")(self.h, C.c_uint32(1 if physics else 0), C.byref(pp), C.byref(rr), C.byref(n), C.byref(st)), "world_view_transforms_landed")
        p = np.ctypeslib.as_array(pp, shape=(n.value, 3)); r = np.ctypeslib.as_array(rr, shape=(n.value, 4))
        p.flags.writeable = False; r.flags.writeable = False
        return p, r, st.value

    def velocities_view(self):
        """mi_world_view_velocities: (linear [n, 3], angular [n, 3]) as read-only views of the library's pinned rows (product library only)."""
        ll = C.POINTER(C.c_float)(); aa = C.POINTER(C.c_float)(); n = C.c_uint32()
        self.L.check(self.L.fn("world_view_velocities")(self.h, C.byref(ll), C.byref(aa), C.byref(n)), "world_view_velocities")
        l = np.ctypeslib.as_array(ll, shape=(n.value, 3)); a = np.ctypeslib.as_array(aa, shape=(n.value, 3))
        l.flags.writeable = False; a.flags.writeable = False
        return l, a

    def pose_stream_stats(self):
        """mi_debug_pose_stream_stats: (rows enqueued by a step itself, rows enqueued only when asked)."""
        a = C.c_uint32(); d = C.c_uint32()
        self.L.check(self.L.fn("debug_pose_stream_stats")(self.h, C.byref(a), C.byref(d)), "debug_pose_stream_stats")
        return a.value, d.value

    def velocities(self):
        return self._get2("world_get_velocities", 3, 3)

    def mass_properties(self):
        n = self.num_entities()
        im = np.zeros(n, np.float32)
        ii = np.zeros((n, 9), np.float32)
        cog = np.zeros((n, 3), np.float32)
        self.L.check(self.L.fn("world_get_mass_properties")(self.h, _ptr(im), _ptr(ii), _ptr(cog), C.c_uint32(n)), "world_get_mass_properties")
        return im, ii, cog

    def counts(self):
        c = StepCounts()
        self.L.check(self.L.fn("world_get_counts")(self.h, C.byref(c)), "world_get_counts")
        return c.as_dict()

    def solve_launches(self):
        """Contact-solve kernel launches of the last step (product library only; 0 for the oracle)."""
        c = StepCounts()
        self.L.check(self.L.fn("world_get_counts")(self.h, C.byref(c)), "world_get_counts")
        return c.reserved

    def contacts(self):
        n = C.c_uint32(0)
        self.L.check(self.L.fn("world_get_contacts")(self.h, None, C.c_uint32(0), C.byref(n)), "world_get_contacts")
        out = np.zeros(n.value, dtype=contact_dtype)
        if n.value:
            self.L.check(self.L.fn("world_get_contacts")(self.h, _ptr(out), C.c_uint32(n.value), C.byref(n)), "world_get_contacts")
        return out

    # --- sharded world (include/mi_shard.h)
    def shard_enable(self, desc):
        self.L.check(self.L.fn("world_shard_enable")(self.h, C.byref(desc)), "world_shard_enable")

    def shard_neighbours(self):
        out = np.zeros(8, np.uint32); n = C.c_uint32()
        self.L.check(self.L.fn("world_shard_neighbours")(self.h, _ptr(out), C.byref(n)), "world_shard_neighbours")
        return [int(x) for x in out[: n.value]]

    def shard_counts(self):
        b = C.c_uint32(); m = C.c_uint32(); c = C.c_uint32()
        self.L.check(self.L.fn("world_shard_counts")(self.h, C.byref(b), C.byref(m), C.byref(c)), "world_shard_counts")
        return {"owned_bodies": b.value, "owned_manifolds": m.value, "owned_contacts": c.value}

    def shard_owned_entities(self):
        n = C.c_uint32()
        self.L.check(self.L.fn("world_shard_owned_entities")(self.h, None, C.c_uint32(0), C.byref(n)), "world_shard_owned_entities")
        out = np.zeros(max(n.value, 1), np.uint32)
        self.L.check(self.L.fn("world_shard_owned_entities")(self.h, _ptr(out), C.c_uint32(len(out)), C.byref(n)), "world_shard_owned_entities")
        return out[: n.value]

    def shard_histogram(self, axis, lo, hi, bins):
        """Owned bodies of the last step per bin of [lo, hi) along x (axis 0) or z (axis 1)."""
        out = np.zeros(bins, np.uint32)
        self.L.check(self.L.fn("world_shard_histogram")(self.h, C.c_uint32(axis), C.c_float(lo), C.c_float(hi), C.c_uint32(bins), _ptr(out)), "world_shard_histogram")
        return out

    def shard_get_borders(self, tiles_x, tiles_z):
        bx = np.zeros(max(tiles_x - 1, 1), np.float32); bz = np.zeros(max(tiles_z - 1, 1), np.float32)
        self.L.check(self.L.fn("world_shard_get_borders")(self.h, _ptr(bx), _ptr(bz)), "world_shard_get_borders")
        return bx[: tiles_x - 1], bz[: tiles_z - 1]

    def shard_set_borders(self, borders_x=None, borders_z=None):
        """New interior tile borders (None = unchanged), in force after the next internal step's exchange; identical on every rank."""
        bx = None if borders_x is None else np.ascontiguousarray(borders_x, np.float32)
        bz = None if borders_z is None else np.ascontiguousarray(borders_z, np.float32)
        self.L.check(self.L.fn("world_shard_set_borders")(self.h, _ptr(bx) if bx is not None and len(bx) else None, _ptr(bz) if bz is not None and len(bz) else None), "world_shard_set_borders")

    def shard_allreduce_u64(self, values):
        """Sum over all ranks through the library transport (one ncclAllReduce on the world's stream)."""
        v = np.ascontiguousarray(values, np.uint64).copy()
        self.L.check(self.L.fn("world_shard_allreduce_u64")(self.h, _ptr(v), C.c_uint32(len(v))), "world_shard_allreduce_u64")
        return v

    def shard_rebalance(self, bins=256):
        self.L.check(self.L.fn("world_shard_rebalance")(self.h, C.c_uint32(bins)), "world_shard_rebalance")

    def shard_message_bytes(self):
        n = C.c_uint64()
        self.L.check(self.L.fn("world_shard_message_bytes")(self.h, C.byref(n)), "world_shard_message_bytes")
        return n.value

    def shard_detach_rccl(self):
        self.L.check(self.L.fn("world_shard_detach_rccl")(self.h), "world_shard_detach_rccl")

    def shard_export(self, slot):
        """The message for neighbour slot `slot` after the last internal step: float32 array, record 0 = (count, ...)."""
        out = np.zeros(self.shard_message_bytes() // 4, np.float32)
        self.L.check(self.L.fn("world_shard_export")(self.h, C.c_uint32(slot), _ptr(out)), "world_shard_export")
        return out

    def shard_import(self, message):
        m = np.ascontiguousarray(message, np.float32)
        self.L.check(self.L.fn("world_shard_import")(self.h, _ptr(m)), "world_shard_import")

    def shard_axis_sums(self):
        """This rank's centre statistics of the last internal step (9 uint64: S1[3], S2lo[3], S2hi[3]); summed over all ranks they give
        the global sweep axis (include/mi_shard.h "Global sweep axis")."""
        out = np.zeros(9, np.uint64)
        self.L.check(self.L.fn("world_shard_axis_sums")(self.h, _ptr(out)), "world_shard_axis_sums")
        return out

    def shard_set_axis_sums(self, global_sums):
        v = np.ascontiguousarray(global_sums, np.uint64)
        assert len(v) == 9
        self.L.check(self.L.fn("world_shard_set_axis_sums")(self.h, _ptr(v)), "world_shard_set_axis_sums")

    def shard_exchange_stats(self, reset=False):
        st = ShardExchangeStats()
        self.L.check(self.L.fn("world_shard_exchange_stats")(self.h, C.byref(st), C.c_uint32(1 if reset else 0)), "world_shard_exchange_stats")
        n = st.num_neighbours
        return {"exchanges": st.exchanges, "device_ms_sum": st.device_ms_sum, "message_bytes": st.message_bytes, "num_neighbours": n,
                "library_transport": bool(st.library_transport), "neighbour_rank": list(st.neighbour_rank[:n]), "records_last": list(st.records_last[:n]),
                "records_sum": list(st.records_sum[:n]), "owned_bodies": st.owned_bodies, "ghost_bodies": st.ghost_bodies,
                "sweep_exchanges": st.sweep_exchanges, "sweep_message_bytes": st.sweep_message_bytes, "sweep_records_last": list(st.sweep_records_last[:n]),
                "message_records_last": list(st.message_records_last[:n]), "message_bytes_sum": st.message_bytes_sum}

    # --- exact seam (include/mi_shard.h "Exact seam")
    def set_seam_tiling(self, desc):
        """A single world orders its contact solve the way the exact-seam sharded worlds of this tiling do (None: plain schedule)."""
        self.L.check(self.L.fn("world_set_seam_tiling")(self.h, C.byref(desc) if desc is not None else None), "world_set_seam_tiling")

    def shard_set_exact_seam(self, enable=True, exchange=None):
        """`exchange(sweep)` is called after every sweep of every internal step (caller's transport); it returns None / 0, or an error code."""
        if exchange is None:
            self._sweep_cb = None
            cb = C.cast(None, SWEEP_FN)
        else:
            def tramp(user, world, sweep):
                try:
                    rc = exchange(int(sweep))
                    return 0 if rc is None else int(rc)
                except BaseException as e:   # an exception cannot cross the C frames: remember it, fail the step
                    self._sweep_error = e
                    return -1
            cb = SWEEP_FN(tramp)
            self._sweep_cb = cb          # keep the trampoline alive as long as the world may call it
        self.L.check(self.L.fn("world_shard_set_exact_seam")(self.h, C.c_uint32(1 if enable else 0), cb, None), "world_shard_set_exact_seam")

    def shard_sweep_message_bytes(self):
        n = C.c_uint64()
        self.L.check(self.L.fn("world_shard_sweep_message_bytes")(self.h, C.byref(n)), "world_shard_sweep_message_bytes")
        return n.value

    def shard_export_sweep(self, slot):
        """The used prefix of the sweep message for neighbour `slot`: header (record 0 = count) + count records of 8 floats."""
        if getattr(self, "_sweep_buf", None) is None or len(self._sweep_buf) != self.shard_sweep_message_bytes() // 4:
            self._sweep_buf = np.zeros(self.shard_sweep_message_bytes() // 4, np.float32)
        out = self._sweep_buf
        self.L.check(self.L.fn("world_shard_export_sweep")(self.h, C.c_uint32(slot), _ptr(out)), "world_shard_export_sweep")
        n = int(out[:1].view(np.uint32)[0])
        return out[: (n + 1) * 8].copy()

    def shard_import_sweep(self, message):
        m = np.ascontiguousarray(message, np.float32)
        self.L.check(self.L.fn("world_shard_import_sweep")(self.h, _ptr(m)), "world_shard_import_sweep")

    def seam_stats(self):
        a = C.c_uint32(); b = C.c_uint32(); c = C.c_uint32()
        self.L.check(self.L.fn("world_seam_stats")(self.h, C.byref(a), C.byref(b), C.byref(c)), "world_seam_stats")
        return {"seam_manifolds": a.value, "seam_colors": b.value, "violations": c.value}

    def shard_attach_rccl(self, unique_id128):
        buf = np.frombuffer(bytes(unique_id128), np.uint8).copy()
        assert len(buf) == 128
        self.L.check(self.L.fn("world_shard_attach_rccl")(self.h, _ptr(buf)), "world_shard_attach_rccl")

    def shard_attach_loopback(self):
        """mi_debug_shard_attach_loopback: the library transport on ONE rank — a one-rank communicator, every neighbour is this rank itself (tests)."""
        self.L.check(self.L.fn("debug_shard_attach_loopback")(self.h), "debug_shard_attach_loopback")

    def shard_peek_received(self, slot, sweep_message=False):
        """mi_debug_shard_peek_received: the message last received in neighbour slot `slot` (library transport)."""
        n = (self.shard_sweep_message_bytes() if sweep_message else self.shard_message_bytes()) // 4
        out = np.zeros(n, np.float32)
        self.L.check(self.L.fn("debug_shard_peek_received")(self.h, C.c_uint32(slot), C.c_uint32(1 if sweep_message else 0), _ptr(out)), "debug_shard_peek_received")
        return out

    # --- ghost-region exchange (13 floats per body: pos3, rot4, lin3, ang3)
    def get_body_states(self, entities):
        ents = np.ascontiguousarray(entities, dtype=np.uint32)
        out = np.zeros((len(ents), 13), np.float32)
        self.L.check(self.L.fn("world_get_body_states")(self.h, C.c_uint32(len(ents)), _ptr(ents), _ptr(out)), "world_get_body_states")
        return out

    def set_body_states(self, entities, states):
        ents = np.ascontiguousarray(entities, dtype=np.uint32)
        st = np.ascontiguousarray(states, dtype=np.float32).reshape(len(ents), 13)
        self.L.check(self.L.fn("world_set_body_states")(self.h, C.c_uint32(len(ents)), _ptr(ents), _ptr(st)), "world_set_body_states")

    def entities_to_bodies(self, entities):
        ents = np.ascontiguousarray(entities, dtype=np.uint32)
        out = np.zeros(len(ents), np.uint32)
        self.L.check(self.L.fn("world_entities_to_bodies")(self.h, C.c_uint32(len(ents)), _ptr(ents), _ptr(out)), "world_entities_to_bodies")
        return out

    def get_body_states_device(self, n, body_ids_ptr, out_ptr):
        self.L.check(self.L.fn("world_get_body_states_device")(self.h, C.c_uint32(n), C.c_void_p(body_ids_ptr), C.c_void_p(out_ptr)),
                     "world_get_body_states_device")

    def set_body_states_device(self, n, body_ids_ptr, in_ptr):
        self.L.check(self.L.fn("world_set_body_states_device")(self.h, C.c_uint32(n), C.c_void_p(body_ids_ptr), C.c_void_p(in_ptr)),
                     "world_set_body_states_device")

    def get_body_states_device_async(self, n, body_ids_ptr, out_ptr):
        self.L.check(self.L.fn("world_get_body_states_device_async")(self.h, C.c_uint32(n), C.c_void_p(body_ids_ptr), C.c_void_p(out_ptr)),
                     "world_get_body_states_device_async")

    def set_body_states_device_async(self, n, body_ids_ptr, in_ptr):
        self.L.check(self.L.fn("world_set_body_states_device_async")(self.h, C.c_uint32(n), C.c_void_p(body_ids_ptr), C.c_void_p(in_ptr)),
                     "world_set_body_states_device_async")

    def stream_ptr(self):
        """The world's hipStream_t as an integer (for torch.cuda.ExternalStream)."""
        out = C.c_void_p()
        self.L.check(self.L.fn("world_get_stream")(self.h, C.byref(out)), "world_get_stream")
        return out.value or 0

    SOLVER_KERNELS = ("k_contact_solve", "k_contact_solve_flow", "k_contact_solve_persist", "k_solve_flow_islands", "k_contact_solve_persist", "k_contact_solve_persist")

    def step_graph_stats(self):
        """mi_debug_step_graph_stats: (enabled, steps replayed as a HIP graph, graphs captured, speculative steps launched plainly)."""
        out = (C.c_uint32 * 4)()
        self.L.check(self.L.fn("debug_step_graph_stats")(self.h, out), "debug_step_graph_stats")
        return tuple(int(x) for x in out)

    def solver_kind(self):
        """mi_world_get_solver_kind: 0 per-colour launches, 1 flow, 2 persistent, 3 flow + joint islands, 4 persistent, XCD-partitioned, 5 persistent, all tiles on one XCD (small piles)."""
        k = C.c_uint32()
        self.L.check(self.L.fn("world_get_solver_kind")(self.h, C.byref(k)), "world_get_solver_kind")
        return k.value

    def solver_kernel(self):
        """Name of the contact-solver kernel the last internal step ran."""
        return self.SOLVER_KERNELS[self.solver_kind()]

    def accumulated_stage_times(self, reset=False):
        """(sum of per-stage device ms, steps, contact updates) since the last reset — product library only."""
        t = StageTimes(); n = C.c_uint32(0); u = C.c_uint64(0)
        self.L.check(self.L.fn("world_get_accumulated_stage_times")(self.h, C.byref(t), C.byref(n), C.byref(u), C.c_uint32(1 if reset else 0)), "world_get_accumulated_stage_times")
        return t.as_dict(), n.value, u.value

    def set_stage_timing(self, level=1):
        """0 / False: nothing is timed (default); 1 / True: every stage; 2: the whole step and the solve stage only (mi_physics.h)."""
        self.L.check(self.L.fn("world_set_stage_timing")(self.h, C.c_uint32(int(level))), "world_set_stage_timing")

    def stage_times(self):
        t = StageTimes()
        self.L.check(self.L.fn("world_get_stage_times")(self.h, C.byref(t)), "world_get_stage_times")
        return t.as_dict()

    def aabbs(self):
        nc = self.counts()["num_colliders"]
        out = np.zeros((nc, 6), np.float32)
        self.L.check(self.L.fn("world_get_aabbs")(self.h, _ptr(out), C.c_uint32(nc)), "world_get_aabbs")
        return out

    def broadphase_pairs(self):
        n = C.c_uint32(0)
        self.L.check(self.L.fn("world_get_broadphase_pairs")(self.h, None, C.c_uint32(0), C.byref(n)), "world_get_broadphase_pairs")
        out = np.zeros((n.value, 2), np.uint32)
        if n.value:
            self.L.check(self.L.fn("world_get_broadphase_pairs")(self.h, _ptr(out), C.c_uint32(n.value), C.byref(n)), "world_get_broadphase_pairs")
        return out
