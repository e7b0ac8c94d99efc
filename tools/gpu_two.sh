#!/bin/bash
# dev helper: colour histogram of the bench scene's schedule
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/col.py <<'PY'
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
for name, make, warm in (("pile", lambda: scenes.obb_pile(128, 16, 128), 290), ("mixed", lambda: scenes.mixed_stack(64, 16, 64), 290)):
    sc = make(); w = sc.populate(mi.create_world(0)); s = sc.settings()
    w.step_fixed(s, sc.dt, warm)
    nm = w.counts()["num_collisions"]
    colors = np.zeros(nm, np.uint32)
    w.L.check(w.L.fn("world_get_manifold_colors")(w.h, colors.ctypes.data_as(C.c_void_p), C.c_uint32(nm)), "colors")
    print(name, nm, np.bincount(colors).tolist(), flush=True)
PY
timeout 200 python /tmp/col.py 2>&1 | tail -2
