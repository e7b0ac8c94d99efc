#!/bin/bash
# dev helper: tiny scene, both solver paths against the oracle, no core dumps
ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
cat > /tmp/dbg.py <<'PY'
import sys, os
sys.path.insert(0, ".")
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
import oracle
sc = scenes.obb_pile(6, 3, 6, spacing=1.0)
g = sc.populate(mi.create_world(0)); o = sc.populate(oracle.create_world(oracle.ORDER_CANONICAL))
s = sc.settings()
for i in range(int(os.environ.get("STEPS", "40"))):
    g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
    if g.counts() != o.counts(): print("counts differ at", i, g.counts(), o.counts()); break
pg, qg = g.physics_transforms(); po, qo = o.physics_transforms()
print(os.environ.get("MI_SOLVER", "flow"), "contacts", g.counts()["num_contacts"], "bit-exact", pg.tobytes() == po.tobytes() and qg.tobytes() == qo.tobytes(), "maxdiff", np.abs(pg - po).max())
PY
MI_SOLVER=launch timeout 120 python /tmp/dbg.py 2>&1 | tail -3 | tee gpurun_out/dbg_launch.log
timeout 120 python /tmp/dbg.py 2>&1 | tail -3 | tee gpurun_out/dbg_flow.log
