#!/bin/bash
# round 4: block solver debugging — small parity cases with per-step block statistics, then growing piles under a time limit
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "other_contact_solvers or block_solver" > gpurun_out/r4b_pytest.log 2>&1
tail -30 gpurun_out/r4b_pytest.log
cat > /tmp/pile.py <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes, capi
import torch
torch.cuda.init()
nx, ny, nz, steps = map(int, sys.argv[1:5])
sc = scenes.obb_pile(nx, ny, nz)
w = sc.populate(mi.create_world(0))
s = sc.settings()
t0 = time.time()
for i in range(steps):
    w.step_fixed(s, sc.dt, 1)
    if time.time() - t0 > 100: print("too slow: stopping at step", i, flush=True); break
t1 = time.time()
import hashlib
p, q = w.physics_transforms()
print("pile", nx, ny, nz, "steps", i + 1, "s/step", (t1 - t0) / (i + 1), "kind", w.solver_kind(), "modes", w.step_mode_stats(), "counts", w.counts(), "sha", hashlib.sha1(p.tobytes() + q.tobytes()).hexdigest()[:12], w.block_stats(), flush=True)
PY
for sz in "16 8 16 150" "32 8 32 200" "64 16 64 260"; do
  MI_BLOCK_DEBUG=1 timeout 200 python /tmp/pile.py $sz > gpurun_out/r4b_pile_blocks_$(echo $sz | tr ' ' _).log 2>&1; echo "rc=$?" >> gpurun_out/r4b_pile_blocks_$(echo $sz | tr ' ' _).log
  MI_SOLVER=persist timeout 200 python /tmp/pile.py $sz > gpurun_out/r4b_pile_persist_$(echo $sz | tr ' ' _).log 2>&1
  tail -2 gpurun_out/r4b_pile_blocks_$(echo $sz | tr ' ' _).log; tail -1 gpurun_out/r4b_pile_persist_$(echo $sz | tr ' ' _).log
done
