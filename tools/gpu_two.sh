#!/bin/bash
# scratch: cfg1 full size, step by step, new vs HEAD build, speculative vs synchronous
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/dbg.py <<'PY'
import sys, os
sys.path.insert(0, ".")
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.sphere_drop(16)
w = sc.populate(mi.create_world(0)); s = sc.settings()
last = None
for i in range(150):
    try:
        w.step_fixed(s, sc.dt, 1)
    except Exception as e:
        print("FAIL at step", i, "last counts", last, str(e)[:200]); break
    last = w.counts()
else:
    print("ok", last, w.step_mode_stats())
PY
for lib in ""; do for a in 1 0; do
echo "lib=$lib async=$a"; MI_DEBUG_SYNC=1 MI_PHYSICS_LIB=$lib MI_ASYNC=$a timeout 120 python /tmp/dbg.py 2>&1 | tail -6
done; done 2>&1 | tee gpurun_out/two.log
