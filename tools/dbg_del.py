import sys, numpy as np
sys.path.insert(0, '.')
import d3d12renderer_amd as mi, oracle
from d3d12renderer_amd import scenes
sc = scenes.shape_zoo()
g = sc.populate(mi.create_world(0)); o = sc.populate(oracle.create_world(oracle.ORDER_CANONICAL))
s = sc.settings()
victims = [3, 77, 143, 10, 11, 142]
schedule = {15 + 12 * k: v for k, v in enumerate(victims)}
for i in range(60):
    if i in schedule:
        for w in (g, o): w.destroy_entity(schedule[i])
        for name in ("physics_transforms", "velocities", "transforms", "mass_properties"):
            a = getattr(g, name)(); b = getattr(o, name)()
            for k,(x,y) in enumerate(zip(a,b)):
                if x.tobytes()!=y.tobytes():
                    bad = np.where((x!=y).reshape(len(x),-1).any(1))[0]; print("after destroy at step", i, name, k, "bad entities", bad[:10], x[bad[:3]], y[bad[:3]])
    g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
    for name in ("physics_transforms", "velocities", "transforms"):
        a = getattr(g, name)(); b = getattr(o, name)()
        for k,(x,y) in enumerate(zip(a,b)):
            if x.tobytes()!=y.tobytes():
                bad = np.where((x!=y).reshape(len(x),-1).any(1))[0]; print("step", i, name, k, "bad entities", bad[:10], np.abs(x-y).max()); 
    if g.counts()!=o.counts(): print("counts", i, g.counts(), o.counts())
