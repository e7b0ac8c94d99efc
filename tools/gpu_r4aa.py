import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI_SOLVER"] = "blocks"
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.obb_pile(14, 8, 14, spacing=1.05)
w = sc.populate(mi.create_world(0)); s = sc.settings()
kinds = []
for i in range(75):
    w.step_fixed(s, sc.dt, 1)
    kinds.append(w.solver_kind())
    if i % 10 == 9 or kinds[-1] != 6: print(i, kinds[-1], w.step_mode_stats(), w.block_stats(), w.counts()["num_collisions"])
print(kinds)
