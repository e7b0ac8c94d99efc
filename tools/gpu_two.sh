#!/bin/bash
# dev helper: XCD-local body fraction of a few scenes
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/xs.py <<'PY'
import sys
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
for name, make in (("mixed", lambda: scenes.mixed_stack(64, 16, 64)), ("terrain", lambda: scenes.terrain_big())):
    sc = make(); w = sc.populate(mi.create_world(0)); s = sc.settings()
    print(name, flush=True)
    w.step_fixed(s, sc.dt, 300)
PY
MI_XCD_STATS=1 timeout 300 python /tmp/xs.py 2>&1 | grep -E "^mixed|^terrain|step 300|step 250" | cut -c1-200
