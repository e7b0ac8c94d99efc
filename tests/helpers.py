import numpy as np
from d3d12renderer_amd import capi, scenes


def single_body_scene(ctype, shape, pos=(0, 0, 0), rot=(0, 0, 0, 1), ground=True, density=1.0, friction=0.5, restitution=0.1,
                      lin_damping=0.4, ang_damping=0.4, iterations=30):
    e = scenes.make_entities(1)
    e["position"][0] = pos
    e["rotation"][0] = rot
    e["linear_damping"] = lin_damping
    e["angular_damping"] = ang_damping
    c = scenes.make_colliders(1, ctype, restitution=restitution, friction=friction, density=density)
    c["shape"][0, :len(shape)] = shape
    ents, cols, cent = [e], [c], [0]
    if ground:
        ge, gc = scenes._ground(100.0)
        ents.append(ge); cols.append(gc); cent.append(1)
    return scenes.Scene("single", np.concatenate(ents), np.asarray(cent, np.uint32), np.concatenate(cols), iterations)


def two_body_scene(descs, iterations=30, gravity=0.0):
    """descs: list of (ctype, shape, pos, rot, kind)."""
    n = len(descs)
    e = scenes.make_entities(n)
    c = scenes.make_colliders(n, 0)
    for i, (ctype, shape, pos, rot, kind) in enumerate(descs):
        e["position"][i] = pos
        e["rotation"][i] = rot
        e["kind"][i] = kind
        e["gravity_factor"][i] = gravity
        c["type"][i] = ctype
        c["shape"][i, :len(shape)] = shape
    return scenes.Scene("pair", e, np.arange(n, dtype=np.uint32), c, iterations)


def contact_set(contacts):
    """Order-independent, bit-exact representation of a contact list."""
    rows = []
    for k in contacts:
        rows.append((int(k["collider_a"]), int(k["collider_b"]), k["point"].tobytes(), k["penetration_depth"].tobytes(),
                     k["normal"].tobytes(), int(k["friction_restitution"]), int(k["body_a"]), int(k["body_b"])))
    return sorted(rows)
