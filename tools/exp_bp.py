"""dev experiment: broad-phase stage time vs. pile state"""
import sys, json
sys.path.insert(0, ".")
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.obb_pile(128, 16, 128)
w = sc.populate(mi.create_world(0)); s = sc.settings()
w.set_stage_timing(True)
done = 0
for target in (240, 400, 600, 900, 1200, 1500):
    w.step_fixed(s, sc.dt, target - done - 5); done = target
    acc = {}
    for _ in range(5):
        w.step_fixed(s, sc.dt, 1)
        for k, v in w.stage_times().items(): acc[k] = acc.get(k, 0) + v / 5
    c = w.counts()
    p, _ = w.physics_transforms()
    y = p[:sc.num_bodies, 1]
    print(target, "bp %.3f np %.3f" % (acc["broadphase"], acc["narrowphase"]), "overlaps", c["num_broadphase_overlaps"], "manifolds", c["num_collisions"], "y max %.1f mean %.2f" % (y.max(), y.mean()), flush=True)
