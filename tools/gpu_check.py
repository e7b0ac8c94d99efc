"""Quick GPU-vs-oracle parity report (development aid; the real checks live in tests/)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
import oracle


def compare(scene, steps, every=10):
    g = scene.populate(mi.create_world())
    o = scene.populate(oracle.create_world(oracle.ORDER_CANONICAL))
    s = scene.settings()
    ok = True
    for i in range(steps):
        g.step_fixed(s, scene.dt, 1)
        o.step_fixed(s, scene.dt, 1)
        cg, co = g.counts(), o.counts()
        pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
        vg, wg = g.velocities(); vo, wo = o.velocities()
        dp = np.abs(pg - po).max(); dq = np.abs(qg - qo).max(); dv = np.abs(vg - vo).max()
        same = cg == co
        if i % every == 0 or not same or dp > 0:
            print(f"step {i}: counts_equal={same} dpos={dp:.3e} dquat={dq:.3e} dvel={dv:.3e} gpu={cg}")
        if not same:
            print("   oracle", co)
            ok = False
            break
        if dp > 1e-3:
            ok = False
            break
    print(scene.name, "PARITY", "OK" if ok else "FAIL", g.stage_times())
    return ok


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg1", "cfg2", "cfg3"]
    if "cfg1" in which: compare(scenes.sphere_drop(8), 150)
    if "cfg2" in which: compare(scenes.mixed_stack(8, 4, 8), 120)
    if "cfg3" in which: compare(scenes.obb_pile(8, 4, 8), 120)
