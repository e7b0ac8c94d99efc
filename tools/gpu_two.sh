#!/bin/bash
# dev helper: the solver-variant parity tests on the GPU box + the 1 M-body pile
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "other_contact_solvers or full_size or retry or bench_size" 2>&1 | tail -5
cat > /tmp/big.py <<'PY'
import sys, time, hashlib
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.obb_pile(256, 16, 256)
w = sc.populate(mi.create_world(0)); s = sc.settings()
w.step_fixed(s, sc.dt, 240)
t0 = time.perf_counter(); w.step_fixed(s, sc.dt, 30); dt = (time.perf_counter() - t0) / 30
st = w.stage_times()
print(sys.argv[1], "1M pile", round(1 / dt, 1), "steps/s solve", round(st["solve"], 3), "total", round(st["total"], 3), w.solver_kernel(), w.solver_kind(), w.counts()["num_contacts"], w.step_mode_stats(),
      hashlib.sha1(w.physics_transforms()[0].tobytes()).hexdigest()[:12], flush=True)
PY
timeout 300 python /tmp/big.py default 2>&1 | tail -1
MI_SOLVER=flow timeout 300 python /tmp/big.py flow 2>&1 | tail -1
