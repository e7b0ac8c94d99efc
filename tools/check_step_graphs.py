"""A graph-replaying world against a plainly launching world of the same scene, bit for bit, every step (run WITHOUT torch: the
library then binds the system HIP runtime, where step graphs are enabled; under the 7.0.x runtime PyTorch bundles they are off).
Prints one JSON line; used by tests/test_gpu_step_graphs.py and as a reproducer (WITH_TORCH=1 MI_GRAPH=force shows the divergence)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.init()
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes

out = {"scenes": {}}
for name, make, steps in (("spheres", lambda: scenes.sphere_drop(10), 260), ("boxes", lambda: scenes.obb_pile(12, 6, 12, spacing=1.05), 200), ("all shapes", lambda: scenes.shape_zoo(8, 5, 8), 200),
                          ("ragdolls", lambda: scenes.ragdolls(4, 4), 260), ("vehicles", lambda: scenes.vehicles(3, 2), 200), ("terrain", lambda: scenes.terrain_field(8, 2, 8), 260)):
    sc = make()
    a = sc.populate(mi.create_world(0))
    outer = os.environ.get("MI_GRAPH")
    os.environ["MI_GRAPH"] = "0"
    b = sc.populate(mi.create_world(0))
    if outer is None: del os.environ["MI_GRAPH"]
    else: os.environ["MI_GRAPH"] = outer
    s = sc.settings()
    a.set_stage_timing(2); b.set_stage_timing(2)      # the whole step and the solve stage timed (recorded events inside a graph): they must survive a replay
    first_bad = None
    for i in range(steps):
        a.step_fixed(s, sc.dt, 1); b.step_fixed(s, sc.dt, 1)
        va, vb = a.velocities(), b.velocities()
        if a.counts() != b.counts() or a.physics_transforms()[0].tobytes() != b.physics_transforms()[0].tobytes() or va[0].tobytes() != vb[0].tobytes() or va[1].tobytes() != vb[1].tobytes():
            first_bad = i; break
    ta = a.stage_times()
    out["scenes"][name] = {"first_mismatch": first_bad, "graph_stats": a.step_graph_stats(), "plain_stats": b.step_graph_stats(), "steps": steps,
                           "times_ok": bool(ta["total"] > 0 and ta["solve"] > 0)}
    a.close(); b.close()
# what a replay saves: a small pile, graphs on / off
outer = os.environ.get("MI_GRAPH")
for label, env in (("graph", None), ("plain", "0")):
    if env is not None: os.environ["MI_GRAPH"] = env
    sc = scenes.sphere_drop(16); w = sc.populate(mi.create_world(0)); s = sc.settings()
    w.step_fixed(s, sc.dt, 500); w.counts()
    t0 = time.perf_counter(); w.step_fixed(s, sc.dt, 300); w.counts()
    out["cfg1_ms_per_step_" + label] = (time.perf_counter() - t0) / 300 * 1e3
    w.close()
    if env is not None:
        if outer is None: del os.environ["MI_GRAPH"]
        else: os.environ["MI_GRAPH"] = outer
print(json.dumps(out))
