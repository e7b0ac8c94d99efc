"""The headline workload THROUGH THE HOST BOUNDARY: what a caller that wants the poses in host memory after every step gets
(physicsStep + reading transform_component of every entity = mi_world_step + mi_world_get_transforms / mi_world_view_transforms), next to
the resident rate bench.py reports.  The pile is still compacting at this point of its life (contacts and step time grow by the hundred
steps), so the variants are measured INTERLEAVED, one frame of each in turn: every variant sees the same states."""
import json, os, sys, time, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes

sc = scenes.obb_pile(128, 16, 128, solver_iterations=20)
w = sc.populate(mi.create_world(0)); s = sc.settings()
w.step_fixed(s, sc.dt, 240)
n = w.num_entities(); P = np.zeros((n, 3), np.float32); R = np.zeros((n, 4), np.float32)      # a C++ caller's own buffers, allocated once
get = w.L.fn("world_get_transforms"); pp, rr = P.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p)
stream = os.environ.get("MI_POSE_STREAM") != "0"
variants = {
    "resident_step_fixed": lambda: (w.step_fixed(s, sc.dt, 1), w.solver_kind()),
    "resident_physicsStep": lambda: (w.step(s, sc.dt), w.solver_kind()),
    "step_fixed_plus_physics_transforms": lambda: (w.step_fixed(s, sc.dt, 1), w.physics_transforms()),
    "physicsStep_plus_entity_transforms": lambda: (w.step(s, sc.dt), w.transforms()),
    "physicsStep_plus_entity_transforms_into_reused_buffers": lambda: (w.step(s, sc.dt), get(w.h, pp, rr, C.c_uint32(n))),
    "physicsStep_plus_transforms_plus_velocities": lambda: (w.step(s, sc.dt), w.transforms(), w.velocities()),
}
if stream:
    variants["physicsStep_plus_entity_transforms_viewed_in_pinned_rows"] = lambda: (w.step(s, sc.dt), w.transforms_view())
FRAMES = 4      # consecutive frames of one variant (the second and later ones have their rows enqueued by the step itself), timed from the second on
times = {k: [] for k in variants}
for rnd in range(30):
    for k, f in variants.items():
        f()
        for _ in range(FRAMES - 1):
            t = time.perf_counter(); f(); times[k].append(time.perf_counter() - t)
out = {"workload": "cfg3 obb_pile 128x16x128 (262144 bodies), settled 240 steps; variants interleaved, 30 rounds x 3 timed frames each", "contacts_at_the_end": w.counts()["num_contacts"]}
for k, v in times.items():
    out[k + "_steps_per_s"] = round(1.0 / st.mean(v), 1); out[k + "_median_ms"] = round(1e3 * st.median(v), 4)
base = st.mean(times["resident_physicsStep"])
out["frame_time_over_resident_step"] = {k: round(st.mean(v) / base, 3) for k, v in times.items()}
out["pose_rows_enqueued_by_the_step_itself_and_on_demand"] = w.pose_stream_stats()
p, r = w.transforms()
out["bytes_to_host_per_step"] = {"transforms": int(p.nbytes + r.nbytes), "with_velocities": int(p.nbytes + r.nbytes + sum(a.nbytes for a in w.velocities()))}
print(json.dumps(out))
