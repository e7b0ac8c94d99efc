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
for i in range(steps): w.step_fixed(s, sc.dt, 1)
print("done", w.counts()["num_contacts"], flush=True)
PY
MI_BLOCK_MODE=0x100 MI_BLOCK_WAVES=4 timeout 100 python /tmp/pile.py 32 8 32 60 2>&1 | grep -v amdgpu | cut -c1-420 | tee gpurun_out/r4s.log
