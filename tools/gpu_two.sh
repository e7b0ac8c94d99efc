#!/bin/bash
# dev helper: A/B of stepping options (environment) on the bench scene and a small scene
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/ab.py <<'PY'
import sys, time, hashlib
sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
for name, make, warm, steps in (("pile262144", lambda: scenes.obb_pile(128, 16, 128), 250, 60), ("spheres4096", lambda: scenes.sphere_drop(16), 240, 200), ("mixed65536", lambda: scenes.mixed_stack(64, 16, 64), 240, 60)):
    sc = make(); w = sc.populate(mi.create_world(0)); s = sc.settings()
    w.step_fixed(s, sc.dt, warm)
    t0 = time.perf_counter()
    for _ in range(steps): w.step_fixed(s, sc.dt, 1)
    dt = (time.perf_counter() - t0) / steps
    st = w.stage_times()
    print(sys.argv[1], name, round(dt * 1e3, 4), "ms/step", round(1 / dt, 1), "steps/s; device total", round(st["total"], 4), "solve", round(st["solve"], 4), w.step_mode_stats(), hashlib.sha1(w.physics_transforms()[0].tobytes()).hexdigest()[:10], flush=True)
PY
run() { timeout 300 python /tmp/ab.py "$@" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3; }
run aux
MI_AUX_STREAM=0 run noaux
run aux
MI_AUX_STREAM=0 run noaux
