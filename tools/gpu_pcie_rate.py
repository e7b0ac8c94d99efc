"""The headline workload THROUGH THE HOST BOUNDARY: what a caller that wants the poses in host memory after every step gets
(physicsStep + reading transform_component of every entity = mi_world_step + mi_world_get_transforms / mi_world_view_transforms), next to
the resident rate bench.py reports.  The pile is still compacting at this point of its life (contacts and step time grow by the hundred
steps: 1.0 ms at step 240, 1.5 ms at step 600), so EVERY variant gets its own fresh world, settled 240 steps like bench.py's, and is
timed over the same 60 frames of the pile's life."""
import ctypes as C
import gc, json, os, sys, time, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes

FRAMES = 60
sc = scenes.obb_pile(128, 16, 128, solver_iterations=20)
s = sc.settings()
stream = os.environ.get("MI_POSE_STREAM") != "0"


def run(make_frame):
    w = sc.populate(mi.create_world(0))
    w.step_fixed(s, sc.dt, 240)
    frame = make_frame(w)
    for _ in range(3): frame()          # (the second and later frames have their rows enqueued by the step itself)
    w.counts()
    ts = []
    t0 = time.perf_counter()
    for _ in range(FRAMES):
        t = time.perf_counter(); frame(); ts.append(time.perf_counter() - t)
    el = time.perf_counter() - t0
    res = {"steps_per_s": round(FRAMES / el, 1), "median_ms": round(1e3 * st.median(ts), 4), "contacts_afterwards": w.counts()["num_contacts"], "pose_rows_by_the_step_itself_and_on_demand": w.pose_stream_stats()}
    del w; gc.collect()
    return res


def reused(w):
    n = w.num_entities(); P = np.zeros((n, 3), np.float32); R = np.zeros((n, 4), np.float32)      # a C++ caller's own buffers, allocated once
    get = w.L.fn("world_get_transforms"); pp, rr = P.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p)
    return lambda: (w.step(s, sc.dt), get(w.h, pp, rr, C.c_uint32(n)), P, R)


variants = {
    "resident_step_fixed": lambda w: (lambda: (w.step_fixed(s, sc.dt, 1), w.solver_kind())),
    "resident_physicsStep": lambda w: (lambda: (w.step(s, sc.dt), w.solver_kind())),
    "step_fixed_plus_physics_transforms": lambda w: (lambda: (w.step_fixed(s, sc.dt, 1), w.physics_transforms())),
    "physicsStep_plus_entity_transforms": lambda w: (lambda: (w.step(s, sc.dt), w.transforms())),
    "physicsStep_plus_entity_transforms_into_reused_buffers": reused,
    "physicsStep_plus_transforms_plus_velocities": lambda w: (lambda: (w.step(s, sc.dt), w.transforms(), w.velocities())),
}
if stream:
    variants["physicsStep_plus_entity_transforms_viewed_in_pinned_rows"] = lambda w: (lambda: (w.step(s, sc.dt), w.transforms_view()))
    variants["physicsStep_plus_entity_transforms_viewed_one_frame_behind"] = lambda w: (lambda: (w.step(s, sc.dt), w.transforms_view_landed()))
    variants["physicsStep_plus_transforms_plus_velocities_viewed_in_pinned_rows"] = lambda w: (lambda: (w.step(s, sc.dt), w.transforms_view(), w.velocities_view()))
out = {"workload": f"cfg3 obb_pile 128x16x128 (262144 bodies); every variant on its own world, settled 240 steps, {FRAMES} timed frames", "pose_stream": stream}
for k, v in variants.items():
    out[k] = run(v)
base = out["resident_physicsStep"]["steps_per_s"]
out["frame_time_over_resident_step"] = {k: round(base / v["steps_per_s"], 3) for k, v in out.items() if isinstance(v, dict) and "steps_per_s" in v}
out["bytes_to_host_per_step"] = {"transforms": 7 * 4 * (sc.num_bodies + 5), "with_velocities": 13 * 4 * (sc.num_bodies + 5), "note": "28 B per entity (position 12 B, rotation 16 B), + 24 B with the velocities"}
print(json.dumps(out))
