"""Development: repeat the per-colour fallback scenario of tests/test_gpu_parity.py (MI_FLOW_FAULT) and report stage times that are not positive."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI_SOLVER"] = "flow"; os.environ["MI_FLOW_FAULT"] = "1"
if os.environ.get("WITH_TORCH"):
    import torch  # the bundled HIP 7.0 runtime is then the one in the process
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for rep in range(n):
    sc = scenes.obb_pile(14, 8, 14, spacing=1.05)
    g = sc.populate(mi.create_world(0)); s = sc.settings()
    for i in range(70):
        g.step_fixed(s, sc.dt, 1)
        t = g.stage_times()
        if not (t["total"] > 0 and t["solve"] > 0 and t["broadphase"] == 0):
            bad += 1; print("rep", rep, "step", i, "kind", g.solver_kind(), t, flush=True)
    g.close()
print("bad", bad, "of", n * 70)
