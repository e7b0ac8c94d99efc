#!/bin/bash
# dev helper: soak run of the bench scene (fallbacks, retries, finiteness over 3000 steps)
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/soak.py <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.obb_pile(128, 16, 128)
w = sc.populate(mi.create_world(0)); s = sc.settings()
kinds = set()
t0 = time.perf_counter()
for blk in range(30):
    w.step_fixed(s, sc.dt, 100); kinds.add(w.solver_kind())
dt = (time.perf_counter() - t0) / 3000
p, q = w.physics_transforms()
print("3000 steps", round(dt * 1e3, 3), "ms/step avg; kinds", kinds, "stats", w.step_mode_stats(), "finite", bool(np.isfinite(p).all() and np.isfinite(q).all()), "min y", float(p[:sc.num_bodies, 1].min()), w.counts()["num_contacts"], flush=True)
PY
timeout 600 python /tmp/soak.py 2>&1 | tail -1
