"""Soak: long runs, looking for anything that only shows up rarely — speculation retries, solver fallbacks, a graph replay that
leaves the plain world.  Run WITHOUT torch to have step graphs on (system HIP runtime).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.init()
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes

out = {}
# 1. the bench pile: 6000 steps, retries / fallbacks / finite state
sc = scenes.obb_pile(128, 16, 128); w = sc.populate(mi.create_world(0)); s = sc.settings()
t0 = time.perf_counter(); w.step_fixed(s, sc.dt, int(os.environ.get("PILE_STEPS", "6000"))); c = w.counts()
p, q = w.physics_transforms()
out["pile_262144"] = {"steps_per_s": int(os.environ.get("PILE_STEPS", "6000")) / (time.perf_counter() - t0), "counts": c, "finite": bool(np.isfinite(p).all() and np.isfinite(q).all()),
                      "min_y": float(p[:, 1].min()), "step_mode_stats(total, speculative, retries)": w.step_mode_stats(), "solver_kind": w.solver_kind()}
w.close()
# 2. small scenes, graph-replaying world against a plain one, compared every 50 steps for 5000 steps
for name, make in (("ragdolls", lambda: scenes.ragdolls(6, 6)), ("spheres", lambda: scenes.sphere_drop(12)), ("zones", lambda: scenes.zones(10, 3, 10)), ("vehicles", lambda: scenes.vehicles(4, 4))):
    sc = make(); a = sc.populate(mi.create_world(0))
    outer = os.environ.get("MI_GRAPH"); os.environ["MI_GRAPH"] = "0"; b = sc.populate(mi.create_world(0))
    if outer is None: del os.environ["MI_GRAPH"]
    else: os.environ["MI_GRAPH"] = outer
    s = sc.settings(); bad = None; n = int(os.environ.get("SMALL_STEPS", "5000"))
    for i in range(0, n, 50):
        a.step_fixed(s, sc.dt, 50); b.step_fixed(s, sc.dt, 50)
        if a.counts() != b.counts() or a.physics_transforms()[0].tobytes() != b.physics_transforms()[0].tobytes() or a.velocities()[0].tobytes() != b.velocities()[0].tobytes():
            bad = i + 50; break
    out[name] = {"first_mismatch_by_step": bad, "graph_stats": a.step_graph_stats(), "stats_graph_world": a.step_mode_stats(), "stats_plain_world": b.step_mode_stats()}
    a.close(); b.close()
print(json.dumps(out))
