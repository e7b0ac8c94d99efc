"""The reference's binary entity stream — serializeEntityToMemory / deserializeEntityFromMemory,
src/scene/serialization_binary.cpp:484-497 — for the physics components.

Layout (serialization_binary.cpp:105-133, 450-480): for every type of `serialized_components`, in order, one `bool` (1 byte) and —
if set — the component written by its serializeToMemoryStream: by default `stream.write(component)`, i.e. the raw MSVC x64
struct image with no padding BETWEEN writes (the stream is a byte array) but with the struct's own tail padding.

    tag_component                 char name[16]
    transform_component           trs: quat rotation (16-byte aligned), vec3 position, vec3 scale, 8 B padding      = 48 B
    position / position_rotation / position_scale        (not used by physics entities: written as absent)
    dynamic_transform_component   flag only (lines 152-153)
    mesh / point light / spot light                      (renderer: absent)
    rigid_body_component          localCOG, invMass, invInertia (mat3, column-major), gravityFactor, linear / angular damping,
                                  linear / angular velocity, force / torque accumulator                             = 112 B
    force_field_component         vec3 force
    cloth_component               width, height, gridSizeX, gridSizeY, totalMass, stiffness, damping, gravityFactor (lines 172-199)
    cloth_render_component        flag only
    physics_reference_component   uint32 numColliders, numColliders x collider_union (80 B: 48 B shape union, physics_material
                                  {type, restitution, friction, density}, collider_type u8, objectType u8, objectIndex u16, padding),
                                  newest collider first; uint32 numConstraints, per constraint (newest first): constraint_type i32,
                                  entity A u32, entity B u32, the constraint struct image (lines 204-232)
    terrain_component             (renderer: absent)
    heightmap_collider_component  chunksPerDim u32, chunkSize f32, physics_material (lines 377-393)
    grass / proc placement / water                        (absent)

trigger_component is not part of the list (it holds a std::function): like in the reference, a trigger entity comes back as a plain
collider entity.  Entity handles of constraints are this library's entity ids (the reference writes EnTT identifiers, which are only
meaningful inside the registry that wrote them — it uses the stream for undo / copy inside one editor session).  Padding bytes are
written as zero.  tests/test_scene_formats.py checks these bytes, field by field, against streams written with the reference's own
struct definitions (oracle/_ref).
"""
import struct

import numpy as np

from . import capi
from .scenes import Scene, make_colliders, make_entities

_COMPONENTS = ("tag", "transform", "position", "position_rotation", "position_scale", "dynamic_transform", "mesh", "point_light", "spot_light",
               "rigid_body", "force_field", "cloth", "cloth_render", "physics_reference", "terrain", "heightmap_collider", "grass",
               "proc_placement", "water")            # serialized_components, serialization_binary.cpp:105-133
_UNSUPPORTED = ("position", "position_rotation", "position_scale", "mesh", "point_light", "spot_light", "terrain", "grass", "proc_placement", "water")

transform_image = np.dtype({"names": ["rotation", "position", "scale"], "formats": [("<f4", 4), ("<f4", 3), ("<f4", 3)], "offsets": [0, 16, 28], "itemsize": 48})
rigid_body_image = np.dtype([("local_cog", "<f4", 3), ("inv_mass", "<f4"), ("inv_inertia", "<f4", 9), ("gravity_factor", "<f4"), ("linear_damping", "<f4"),
                             ("angular_damping", "<f4"), ("linear_velocity", "<f4", 3), ("angular_velocity", "<f4", 3), ("force_accumulator", "<f4", 3),
                             ("torque_accumulator", "<f4", 3)])
collider_image = np.dtype({"names": ["shape", "hull_geometry", "material_type", "restitution", "friction", "density", "type", "object_type", "object_index"],
                           "formats": [("<f4", 12), "<u4", "<i4", "<f4", "<f4", "<f4", "u1", "u1", "<u2"],
                           "offsets": [0, 32, 48, 52, 56, 60, 64, 65, 66], "itemsize": 80})
cloth_image = np.dtype([("width", "<f4"), ("height", "<f4"), ("grid_size_x", "<u4"), ("grid_size_y", "<u4"), ("total_mass", "<f4"), ("stiffness", "<f4"),
                        ("damping", "<f4"), ("gravity_factor", "<f4")])
heightmap_image = np.dtype([("chunks_per_dim", "<u4"), ("chunk_size", "<f4"), ("material_type", "<i4"), ("restitution", "<f4"), ("friction", "<f4"), ("density", "<f4")])
assert rigid_body_image.itemsize == 112 and cloth_image.itemsize == 32 and heightmap_image.itemsize == 24
# constraint struct images: the ABI PODs are field-for-field the reference structs; fixed and slider start with a quat, so MSVC pads them to 16 bytes
CONSTRAINT_IMAGE_BYTES = [28, 24, 48, 104, 120, 80]


def _shape_to_image(ctype, shape12):
    """mi_collider_desc::shape -> the 48-byte bounding-volume union (the OBB / hull images start with the quaternion, like the ABI's)."""
    out = np.zeros(12, np.float32)
    n = {capi.SPHERE: 4, capi.CAPSULE: 7, capi.CYLINDER: 7, capi.AABB: 6, capi.OBB: 10, capi.HULL: 7}[int(ctype)]
    out[:n] = shape12[:n]
    return out


def serialize_entity(scene, world, entity, name="e"):
    """serializeEntityToMemory(entity, ...) for entity id `entity` of `scene`, with the current state read back from `world`
    (product, oracle or reference world alike: transforms, velocities, mass properties, constraint PODs)."""
    e = scene.entities[entity]
    kind = int(e["kind"])
    out = bytearray()
    pos, rot = world.transforms()
    lin, ang = world.velocities()
    inv_mass, inv_inertia, cog = world.mass_properties()

    def flag(has):
        out.append(1 if has else 0)
        return has

    flag(True); out += struct.pack("16s", name.encode()[:15])
    t = np.zeros(1, transform_image); t["rotation"] = rot[entity]; t["position"] = pos[entity]; t["scale"] = 1.0
    flag(True); out += t.tobytes()
    flag(False); flag(False); flag(False)
    is_body = kind in (capi.ENTITY_DYNAMIC, capi.ENTITY_KINEMATIC)
    flag(is_body)                                              # dynamic_transform_component comes with the rigid body (scene.h:78-81)
    flag(False); flag(False); flag(False)
    if flag(is_body):
        rb = np.zeros(1, rigid_body_image)
        rb["local_cog"] = cog[entity]; rb["inv_mass"] = inv_mass[entity]; rb["inv_inertia"] = inv_inertia[entity]
        rb["gravity_factor"] = e["gravity_factor"]; rb["linear_damping"] = e["linear_damping"]; rb["angular_damping"] = e["angular_damping"]
        rb["linear_velocity"] = lin[entity]; rb["angular_velocity"] = ang[entity]          # accumulators: zero between steps
        out += rb.tobytes()
    if flag(kind == capi.ENTITY_FORCE_FIELD):
        force = next((f for ent, f in scene.forces if ent == entity), (0.0, 0.0, 0.0))
        out += np.asarray(force, np.float32).tobytes()
    flag(False); flag(False)                                   # cloth, cloth render
    mine = [i for i in range(len(scene.colliders)) if int(scene.collider_entities[i]) == entity]
    edges = _constraint_edges(scene, entity)
    if flag(bool(mine) or bool(edges)):
        out += struct.pack("<I", len(mine))
        for i in reversed(mine):                               # the entity's collider list is newest first (scene.h:52-54)
            c = scene.colliders[i]
            img = np.zeros(1, collider_image)
            img["shape"] = _shape_to_image(c["type"], c["shape"])
            if int(c["type"]) == capi.HULL:
                img["hull_geometry"] = c["hull_geometry"]
            img["material_type"] = -1; img["restitution"] = c["restitution"]; img["friction"] = c["friction"]; img["density"] = c["density"]
            img["type"] = c["type"]
            out += img.tobytes()
        out += struct.pack("<I", len(edges))
        for ctype, cid, ea, eb in edges:
            pod = world.get_constraint(ctype, cid).tobytes()
            out += struct.pack("<iII", ctype, ea, eb) + pod + b"\0" * (CONSTRAINT_IMAGE_BYTES[ctype] - len(pod))
    flag(False); flag(False); flag(False); flag(False); flag(False)
    return bytes(out)


def _constraint_list(scene):
    """(type, id within the type, entity a, entity b) of every constraint of the scene in creation order (Scene.populate's order)."""
    seen = [0] * 6; out = []
    for ctype, ea, eb, *_ in list(scene.constraints) + list(scene.global_constraints):
        out.append((int(ctype), seen[int(ctype)], int(ea), int(eb))); seen[int(ctype)] += 1
    return out


def _constraint_edges(scene, entity):
    """The entity's constraint edge list, newest first (addConstraintEdge prepends, physics.cpp:87-126)."""
    return [c for c in reversed(_constraint_list(scene)) if entity in (c[2], c[3])]


def parse_entity(blob):
    """deserializeEntityFromMemory's walk over one entity's bytes -> dict of numpy records.  Raises ValueError on truncated input, trailing
    bytes, or a component this module cannot size (renderer components)."""
    view = memoryview(blob); off = 0
    out = {}

    def take(n):
        nonlocal off
        if off + n > len(view):
            raise ValueError("truncated entity stream")
        b = bytes(view[off:off + n]); off += n
        return b

    for comp in _COMPONENTS:
        if not take(1)[0]:
            continue
        if comp in _UNSUPPORTED:
            raise ValueError(f"entity stream holds a {comp} component (renderer data): not a physics entity")
        if comp == "tag":
            out["tag"] = take(16).split(b"\0")[0].decode(errors="replace")
        elif comp == "transform":
            out["transform"] = np.frombuffer(take(48), transform_image)[0]
        elif comp in ("dynamic_transform", "cloth_render"):
            out[comp] = True
        elif comp == "rigid_body":
            out["rigid_body"] = np.frombuffer(take(112), rigid_body_image)[0]
        elif comp == "force_field":
            out["force_field"] = np.frombuffer(take(12), "<f4")
        elif comp == "cloth":
            out["cloth"] = np.frombuffer(take(32), cloth_image)[0]
        elif comp == "heightmap_collider":
            out["heightmap_collider"] = np.frombuffer(take(24), heightmap_image)[0]
        elif comp == "physics_reference":
            n = struct.unpack("<I", take(4))[0]
            out["colliders"] = np.frombuffer(take(80 * n), collider_image) if n else np.zeros(0, collider_image)
            m = struct.unpack("<I", take(4))[0]
            cons = []
            for _ in range(m):
                ctype, ea, eb = struct.unpack("<iII", take(12))
                if not 0 <= ctype < 6:
                    raise ValueError("bad constraint type in entity stream")
                img = take(CONSTRAINT_IMAGE_BYTES[ctype])
                dt = capi.CONSTRAINT_DTYPES[ctype]
                cons.append((ctype, ea, eb, np.frombuffer(img[:dt.itemsize], dt)[0]))
            out["constraints"] = cons
    if off != len(view):
        raise ValueError("trailing bytes after the entity stream")      # deserializeEntityFromMemory returns readOffset == size
    return out


def load_entities(blobs, solver_iterations=30, name="binary"):
    """A list of entity streams (entity id = position in the list) -> Scene.  A constraint is listed by both of its entities;
    it is created once, when the entity with the smaller id is read (the reference's deserializer would add it twice if both
    entities were restored — it only ever restores one entity at a time)."""
    parsed = [parse_entity(b) for b in blobs]
    ents = make_entities(len(parsed), capi.ENTITY_STATIC)
    col_ents, cols, forces, constraints = [], [], [], []
    for i, p in enumerate(parsed):
        t = p.get("transform")
        if t is not None:
            ents["position"][i] = t["position"]; ents["rotation"][i] = t["rotation"]
        rb = p.get("rigid_body")
        if rb is not None:
            ents["kind"][i] = capi.ENTITY_KINEMATIC if rb["inv_mass"] == 0.0 and not np.any(rb["inv_inertia"]) else capi.ENTITY_DYNAMIC
            for k in ("gravity_factor", "linear_damping", "angular_damping", "linear_velocity", "angular_velocity"):
                ents[k][i] = rb[k]
        if "force_field" in p:
            ents["kind"][i] = capi.ENTITY_FORCE_FIELD; forces.append((i, tuple(float(x) for x in p["force_field"])))
        for img in reversed(p.get("colliders", [])):           # stored newest first: add oldest first to rebuild the same list
            c = make_colliders(1, int(img["type"]), restitution=float(img["restitution"]), friction=float(img["friction"]), density=float(img["density"]))
            c["shape"][0] = img["shape"]
            if int(img["type"]) == capi.HULL:
                c["hull_geometry"][0] = img["hull_geometry"]
            cols.append(c); col_ents.append(i)
    order = {}
    for i, p in enumerate(parsed):
        for k, (ctype, ea, eb, pod) in enumerate(reversed(p.get("constraints", []))):      # oldest first
            if i == min(ea, eb):
                constraints.append((ctype, ea, eb, np.array([pod], dtype=capi.CONSTRAINT_DTYPES[ctype])))
    colliders = np.concatenate(cols) if cols else make_colliders(0, capi.SPHERE)
    return Scene(name, ents, np.asarray(col_ents, np.uint32), colliders, solver_iterations, constraints=constraints, forces=forces)
