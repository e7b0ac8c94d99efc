"""Development: one fuzz world (tools/gpu_fuzz.py) on the GPU, step by step, with parts of it removed: python tools/exp/fuzz_bisect.py SEED [no-terrain no-zones no-joints no-events no-actions no-hulls]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gpu_fuzz                                   # noqa: E402
import d3d12renderer_amd as mi                    # noqa: E402
from d3d12renderer_amd import capi, scenes        # noqa: E402

seed = int(sys.argv[1]); off = set(sys.argv[2:])
sc, bodies, rng = gpu_fuzz.make_world_description(seed)
events = bool(rng.random() < 0.5) and "no-events" not in off
if "no-terrain" in off and sc.heightmap is not None:
    sc.heightmap = None
    g = scenes.make_entities(1, capi.ENTITY_STATIC); gc = scenes.make_colliders(1, capi.AABB, 0.1, 0.5, 4.0); gc["shape"][0, :6] = (-60.0, -4.0, -60.0, 60.0, 0.0, 60.0)
    sc.collider_entities = np.concatenate([sc.collider_entities, [len(sc.entities)]]).astype(np.uint32); sc.entities = np.concatenate([sc.entities, g]); sc.colliders = np.concatenate([sc.colliders, gc])
if "no-zones" in off:
    z = np.isin(sc.entities["kind"], (capi.ENTITY_TRIGGER, capi.ENTITY_FORCE_FIELD)); sc.entities["kind"][z] = capi.ENTITY_STATIC; sc.forces = []
if "no-joints" in off:
    sc.global_constraints = []
if "no-hulls" in off:
    h = sc.colliders["type"] == capi.HULL; sc.colliders["type"][h] = capi.SPHERE; sc.colliders["shape"][h, :4] = (0, 0, 0, 0.3)
plan = gpu_fuzz.plan_actions(seed, 40)
if "no-actions" in off:
    plan = [(m, 1.0, r, v) for m, u, r, v in plan]
w = sc.populate(mi.create_world(0))
print("populated", flush=True)
r = gpu_fuzz.run_world(w, sc, 40, plan, events, bodies)
print("ok", r[-1][0], flush=True)
