"""Development: a graph-replaying world against a plainly launching world of the same scene, step by step."""
import sys, os; sys.path.insert(0, ".")
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.init()
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.ragdolls(4, 4)
a = sc.populate(mi.create_world(0))
os.environ["MI_GRAPH"] = "0"
b = sc.populate(mi.create_world(0))
s = sc.settings()
for i in range(130):
    a.step_fixed(s, sc.dt, 1); b.step_fixed(s, sc.dt, 1)
    ca, cb = a.contacts(), b.contacts()
    same_c = ca.tobytes() == cb.tobytes()
    va, vb = a.velocities(), b.velocities()
    same_v = va[0].tobytes() == vb[0].tobytes() and va[1].tobytes() == vb[1].tobytes()
    same_p = a.physics_transforms()[0].tobytes() == b.physics_transforms()[0].tobytes()
    if not (same_c and same_v and same_p) or a.counts() != b.counts():
        print("step", i, "contacts equal", same_c, "velocities equal", same_v, "poses equal", same_p, a.counts(), b.counts(), a.step_mode_stats(), b.step_mode_stats())
        d = np.abs(va[0] - vb[0]).max(axis=1); print("bodies with different linear velocity:", np.nonzero(d)[0][:20], d.max())
        break
else:
    print("all equal")
