"""The headline workload THROUGH THE HOST BOUNDARY: what a caller that wants the poses in host memory after every step gets
(physicsStep + reading transform_component of every entity = mi_world_step + mi_world_get_transforms), next to the resident rate bench.py reports."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes

sc = scenes.obb_pile(128, 16, 128, solver_iterations=20)
w = sc.populate(mi.create_world(0)); s = sc.settings()
w.step_fixed(s, sc.dt, 240)
out = {"workload": "cfg3 obb_pile 128x16x128 (262144 bodies), settled 240 steps", "steps": 60}
def timed(f, n=60):
    f(); w.counts()
    t = time.perf_counter()
    for _ in range(n): f()
    w.counts()
    return n / (time.perf_counter() - t)
out["resident_steps_per_s"] = timed(lambda: w.step_fixed(s, sc.dt, 1))
out["step_plus_physics_transforms_steps_per_s"] = timed(lambda: (w.step_fixed(s, sc.dt, 1), w.physics_transforms()))
out["physicsStep_plus_entity_transforms_steps_per_s"] = timed(lambda: (w.step(s, sc.dt), w.transforms()))
out["physicsStep_plus_transforms_plus_velocities_steps_per_s"] = timed(lambda: (w.step(s, sc.dt), w.transforms(), w.velocities()))
import ctypes as C
n = w.num_entities(); P = np.zeros((n, 3), np.float32); R = np.zeros((n, 4), np.float32)      # a C++ caller's own buffers, allocated once
get = w.L.fn("world_get_transforms"); pp, rr = P.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p)
out["physicsStep_plus_entity_transforms_into_reused_buffers_steps_per_s"] = timed(lambda: (w.step(s, sc.dt), get(w.h, pp, rr, C.c_uint32(n))))
p, r = w.transforms()
out["bytes_to_host_per_step"] = {"transforms": int(p.nbytes + r.nbytes), "with_velocities": int(p.nbytes + r.nbytes + sum(a.nbytes for a in w.velocities()))}
print(json.dumps(out))
