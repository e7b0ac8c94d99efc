"""The reference's YAML scene files (`*.sc`) — src/scene/serialization_yaml.cpp:74-233 (component encodings), 364-524 (file
layout) — as the import / export path for physics scenes.

Only what the physics path consumes is read: `Tag`, `Transform` / `Position` / `Position/Rotation` / `Position/Scale`
(src/scene/components.h:29-69), `Rigid body` (Local COG, Inv mass, Inv inertia, Gravity factor, Linear / Angular damping),
`Force field` (Force) and `Colliders` (Type Sphere | Capsule | AABB | OBB with their fields + Restitution / Friction /
Density).  Everything else in a scene file (camera, rendering settings, sun, environment, meshes, lights, cloth) is ignored
on load and written as neutral defaults on save.  Like the reference, hull colliders are not representable (decode returns
false, serialization_yaml.cpp:221-224), cylinders have no case at all (120-155), and constraints are not serialized ("TODO",
462-467).  Mass properties are recomputed from the colliders when they are added, exactly as the reference does on load
(the stored Local COG / Inv mass / Inv inertia only tell a kinematic body — Inv mass 0 — from a dynamic one).
"""
import numpy as np
import yaml

from . import capi
from .scenes import Scene, make_colliders, make_entities

_TYPE_NAMES = ["Sphere", "Capsule", "Cylinder", "AABB", "OBB", "Hull"]          # colliderTypeNames, src/physics/physics.h:72-80


def _f(v):
    return float(np.float32(v))


def _vec(a):
    return [_f(x) for x in a]


def _entity_transform(node):
    pos, rot = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0)
    for key in ("Transform", "Position/Rotation", "Position", "Position/Scale"):
        if key in node:
            t = node[key]
            pos = tuple(t.get("Position", pos))
            rot = tuple(t.get("Rotation", rot))
            break
    return pos, rot


def load_scene(text_or_path, solver_iterations=30, name=None):
    """Parses a reference scene file (path or YAML text) into a `Scene` (entities in file order)."""
    text = text_or_path
    if "\n" not in str(text_or_path) and str(text_or_path).endswith((".sc", ".yaml", ".yml")):
        with open(text_or_path) as fh:
            text = fh.read()
    doc = yaml.safe_load(text)
    if not isinstance(doc, dict) or "Scene" not in doc:
        raise ValueError("not a scene file: no 'Scene' key (deserializeSceneFromYAMLFile returns false)")
    nodes = doc.get("Entities") or []
    ents = make_entities(len(nodes))
    col_ents, cols, forces = [], [], []
    for i, node in enumerate(nodes):
        pos, rot = _entity_transform(node)
        ents["position"][i] = pos; ents["rotation"][i] = rot
        if "Rigid body" in node:
            rb = node["Rigid body"]
            ents["kind"][i] = capi.ENTITY_KINEMATIC if float(rb.get("Inv mass", 1.0)) == 0.0 else capi.ENTITY_DYNAMIC
            ents["gravity_factor"][i] = rb.get("Gravity factor", 1.0)
            ents["linear_damping"][i] = rb.get("Linear damping", 0.4)
            ents["angular_damping"][i] = rb.get("Angular damping", 0.4)
        elif "Force field" in node:
            ents["kind"][i] = capi.ENTITY_FORCE_FIELD
            forces.append((i, tuple(node["Force field"].get("Force", (0, 0, 0)))))
        else:
            ents["kind"][i] = capi.ENTITY_STATIC
        for cn in node.get("Colliders") or []:
            t = cn.get("Type")
            if t not in _TYPE_NAMES:
                raise ValueError(f"entity {i}: unknown collider type {t!r}")
            c = make_colliders(1, _TYPE_NAMES.index(t), cn.get("Restitution", 0.0), cn.get("Friction", 0.0), cn.get("Density", 0.0))
            if t == "Sphere":
                c["shape"][0, :4] = (*cn["Center"], cn["Radius"])
            elif t == "Capsule":
                c["shape"][0, :7] = (*cn["Position A"], *cn["Position B"], cn["Radius"])
            elif t == "AABB":
                c["shape"][0, :6] = (*cn["Min corner"], *cn["Max corner"])
            elif t == "OBB":
                c["shape"][0, :10] = (*cn["Rotation"], *cn["Center"], *cn["Radius"])
            else:       # Cylinder: no decode case in the reference; Hull: decode returns false -> the collider is dropped
                continue
            col_ents.append(i); cols.append(c)
    colliders = np.concatenate(cols) if cols else make_colliders(0, capi.SPHERE)
    sc = Scene(name or str(doc.get("Scene")), ents, np.asarray(col_ents, np.uint32), colliders, solver_iterations, forces=forces)
    sc.tags = [str(n.get("Tag", "")) for n in nodes]
    return sc


def dump_scene(scene, world=None, title="My scene"):
    """The YAML text `serializeSceneToYAMLFile` would write for the physics content of `scene` (mass properties from `world`
    if given).  Raises for content the format cannot hold (hulls, cylinders, constraints, triggers)."""
    if scene.constraints or scene.global_constraints:
        raise ValueError("constraints are not part of the reference's scene format (serialization_yaml.cpp:462-467)")
    mass = world.mass_properties() if world is not None else None
    per_entity = {}
    for e, c in zip(scene.collider_entities, scene.colliders):
        per_entity.setdefault(int(e), []).append(c)
    force_of = {int(e): f for e, f in scene.forces}
    out = []
    for i, e in enumerate(scene.entities):
        kind = int(e["kind"])
        if kind == capi.ENTITY_TRIGGER:
            raise ValueError("trigger components are not part of the reference's scene format")
        tags = getattr(scene, "tags", None)
        n = {"Tag": tags[i] if tags else f"Entity {i}",
             "Transform": {"Position": _vec(e["position"]), "Rotation": _vec(e["rotation"]), "Scale": [1.0, 1.0, 1.0]}}
        if kind in (capi.ENTITY_DYNAMIC, capi.ENTITY_KINEMATIC):
            n["Dynamic"] = True
            inv_mass = 0.0 if kind == capi.ENTITY_KINEMATIC else (_f(mass[0][i]) if mass else 1.0)
            inv_inertia = [0.0] * 9 if kind == capi.ENTITY_KINEMATIC else (_vec(np.asarray(mass[1][i]).reshape(3, 3).T.reshape(-1)) if mass else [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0])
            n["Rigid body"] = {"Local COG": _vec(mass[2][i]) if mass else [0.0, 0.0, 0.0], "Inv mass": inv_mass, "Inv inertia": inv_inertia,
                               "Gravity factor": _f(e["gravity_factor"]), "Linear damping": _f(e["linear_damping"]), "Angular damping": _f(e["angular_damping"])}
        if kind == capi.ENTITY_FORCE_FIELD:
            n["Force field"] = {"Force": _vec(force_of.get(i, (0, 0, 0)))}
        cl = []
        for c in reversed(per_entity.get(i, [])):   # collider_component_iterator walks the entity's list newest first
            t = int(c["type"]); s = c["shape"]
            m = {"Type": _TYPE_NAMES[t]}
            if t == capi.SPHERE: m.update({"Center": _vec(s[0:3]), "Radius": _f(s[3])})
            elif t == capi.CAPSULE: m.update({"Position A": _vec(s[0:3]), "Position B": _vec(s[3:6]), "Radius": _f(s[6])})
            elif t == capi.AABB: m.update({"Min corner": _vec(s[0:3]), "Max corner": _vec(s[3:6])})
            elif t == capi.OBB: m.update({"Center": _vec(s[4:7]), "Radius": _vec(s[7:10]), "Rotation": _vec(s[0:4])})
            else: raise ValueError(f"{_TYPE_NAMES[t]} colliders cannot be written to / read from the reference's scene format")
            m.update({"Restitution": _f(c["restitution"]), "Friction": _f(c["friction"]), "Density": _f(c["density"])})
            cl.append(m)
        if cl:
            n["Colliders"] = cl
        out.append(n)
    doc = {"Scene": title, "Entities": out}
    return yaml.safe_dump(doc, sort_keys=False, default_flow_style=None)


def save_scene(scene, path, world=None, title="My scene"):
    with open(path, "w") as fh:
        fh.write(dump_scene(scene, world, title))
