#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cat > /tmp/pile.py <<'PY'
import sys, time, os, hashlib
sys.path.insert(0, os.getcwd())
import torch; torch.cuda.init()
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
nx, ny, nz, steps = map(int, sys.argv[1:5])
sc = scenes.obb_pile(nx, ny, nz); w = sc.populate(mi.create_world(0)); s = sc.settings()
t0 = time.time()
for i in range(steps): w.step_fixed(s, sc.dt, 1)
p, q = w.physics_transforms()
print(os.environ.get("TAG"), "ms/step", round((time.time() - t0) / steps * 1e3, 3), "kind", w.solver_kind(), "modes", w.step_mode_stats(), "contacts", w.counts()["num_contacts"], "sha", hashlib.sha1(p.tobytes() + q.tobytes()).hexdigest()[:12], flush=True)
PY
for v in vm; do for wv in 4; do for rep in 1 2 3; do TAG=$v-w$wv MI_PHYSICS_LIB=build_exp/libmi_$v.so MI_BLOCK_WAVES=$wv timeout 100 python /tmp/pile.py 32 8 32 200 2>&1 | tail -1; done; done; done | tee gpurun_out/r4v.log
