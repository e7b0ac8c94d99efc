#!/bin/bash
# dev helper: throughput of the batched learning environments (libPhysics-Lib.so) on the GPU box
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/learn.py <<'PY'
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
from d3d12renderer_amd.learning import PhysicsDLL
d = PhysicsDLL(); d.seed(1)
_, _, amin, amax = d.ranges()
out = {}
for n in (1, 256, 4096, 16384):
    d.shutdown()
    d.reset_batch(n)
    rng = np.random.default_rng(0)
    acts = [(rng.uniform(-1, 1, (n, 27)) * 0.15 * (amax - amin)).astype(np.float32) for _ in range(8)]
    for i in range(30): d.step_batch(acts[i % 8])
    t0 = time.perf_counter(); resets = 0
    steps = 100
    for i in range(steps):
        _, _, dn = d.step_batch(acts[i % 8]); resets += int(dn.sum())
    dt = (time.perf_counter() - t0) / steps
    out[n] = dict(ms_per_step=dt * 1e3, env_steps_per_s=n / dt, resets=resets)
    print(n, json.dumps(out[n]), flush=True)
json.dump(out, open("gpurun_out/learning.json", "w"), indent=1)
PY
timeout 900 python /tmp/learn.py 2>&1 | tail -6
