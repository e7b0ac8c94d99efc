"""Development: the scenes of test_gpu_vs_oracle_trajectory_and_contacts in one process, worlds kept alive, first mismatch reported."""
import sys; sys.path.insert(0, ".")
import os
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.init()
import numpy as np, oracle
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
keep = []
only = os.environ.get('ONLY')
for make, steps in [(lambda: scenes.sphere_drop(10), 130), (lambda: scenes.mixed_stack(12, 6, 12), 80), (lambda: scenes.obb_pile(12, 8, 12, spacing=1.1), 80),
                    (lambda: scenes.shape_zoo(8, 5, 8), 150), (lambda: scenes.ragdolls(4, 4), 160), (lambda: scenes.joint_zoo(copies=3), 200), (lambda: scenes.vehicles(3, 2), 160)]:
    sc = make()
    if only and only not in sc.name: continue
    g = sc.populate(mi.create_world(0)); o = sc.populate(oracle.create_world(oracle.ORDER_CANONICAL)); s = sc.settings()
    keep.append(g)
    bad = None
    for i in range(steps):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        a, b = g.counts(), o.counts()
        if i % 20 == 0: g.contacts()
        same = g.physics_transforms()[0].tobytes() == o.physics_transforms()[0].tobytes()
        if a != b or not same:
            bad = (i, a == b, same, g.solver_kind(), g.step_mode_stats()); break
    print(sc.name, "MISMATCH at step %d: counts equal %s, poses equal %s, kind %d, stats %s" % bad if bad else "all equal", flush=True)
