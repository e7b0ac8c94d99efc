#!/bin/bash
# dev helper: quick solver timing at the bench state + small-pile parity for solver switches given in the environment
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
cat > /tmp/ko.py <<'PY'
import sys, time, os, hashlib
sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
import oracle as om
om.build()
tag = sys.argv[1]
sc = scenes.obb_pile(14, 8, 14, spacing=1.05)
g = sc.populate(mi.create_world(0)); o = sc.populate(om.create_world(om.ORDER_CANONICAL)); s = sc.settings()
os.environ["MI_PERSIST_XCD_MIN_SAVED"] = "1"
for i in range(70):
    g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
print(tag, "small parity", g.physics_transforms()[0].tobytes() == o.physics_transforms()[0].tobytes(), g.solver_kind(), g.step_mode_stats(), flush=True)
sc = scenes.obb_pile(128, 16, 128)
w = sc.populate(mi.create_world(0)); s = sc.settings()
w.step_fixed(s, sc.dt, 280)
acc = []; tot = []
for _ in range(20):
    w.step_fixed(s, sc.dt, 1); st = w.stage_times(); acc.append(st["solve"]); tot.append(st["total"])
print(tag, "solve", round(float(np.median(acc)), 4), "total", round(float(np.median(tot)), 4), w.solver_kind(), w.step_mode_stats(), hashlib.sha1(w.physics_transforms()[0].tobytes()).hexdigest()[:12], flush=True)
PY
run() { timeout 300 python /tmp/ko.py "$@" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -2; }
run one
MI_PERSIST_TWO=1 run two
MI_PERSIST_TWO=1 MI_PERSIST_XCD=0 run two-noxcd
