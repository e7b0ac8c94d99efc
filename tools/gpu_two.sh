#!/bin/bash
# dev helper: how many extra colouring rounds does speculation need? (retries vs margin)
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/cm.py <<'PY'
import sys, time
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
for name, make in (("pile", lambda: scenes.obb_pile(128, 16, 128)), ("mixed", lambda: scenes.mixed_stack(64, 16, 64))):
    sc = make(); w = sc.populate(mi.create_world(0)); s = sc.settings()
    w.step_fixed(s, sc.dt, 200); r0 = w.step_mode_stats()[2]
    t0 = time.perf_counter(); w.step_fixed(s, sc.dt, 600); dt = (time.perf_counter() - t0) / 600
    print(sys.argv[1], name, "ms/step", round(dt * 1e3, 4), "retries in 600 steps", w.step_mode_stats()[2] - r0, "(first 200:", r0, ")", flush=True)
PY
for m in 3 2 1 0; do MI_COLOR_MARGIN=$m timeout 300 python /tmp/cm.py margin$m 2>&1 | tail -2; done
