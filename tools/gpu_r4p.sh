#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cat > /tmp/cmp.py <<'PY'
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import torch; torch.cuda.init()
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes, capi
def world(path):
    L = capi.Library(path, prefix="mi_"); desc = capi.WorldDesc(0, 0); h = C.c_void_p()
    L.check(L.fn("world_create")(C.byref(desc), C.byref(h)), "world_create"); return capi.World(L, h)
sc = scenes.obb_pile(32, 8, 32)
a = sc.populate(world("build_exp/libmi_vg.so")); b = sc.populate(world("d3d12renderer_amd/libmi_physics.so"))
s = sc.settings()
for i in range(200):
    a.step_fixed(s, sc.dt, 1); b.step_fixed(s, sc.dt, 1)
    va = np.concatenate(a.velocities(), axis=1); vb = np.concatenate(b.velocities(), axis=1)
    if va.tobytes() != vb.tobytes():
        bad = np.where((va != vb).any(axis=1))[0]
        pa = a.physics_transforms()[0]
        print("step", i, "kinds", a.solver_kind(), b.solver_kind(), "bodies that differ", len(bad), "of", len(va), "counts", a.counts()["num_contacts"], b.counts()["num_contacts"])
        print("first ids", bad[:20])
        d = np.abs(va[bad] - vb[bad]).max(axis=1)
        print("max abs diff", d.max(), "median", np.median(d))
        print("positions of first", pa[bad[:8]])
        print("block stats", b.block_stats())
        break
else:
    print("identical for 200 steps")
PY
for rep in 1 2 3; do MI_BLOCK_WAVES=4 timeout 120 python /tmp/cmp.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r4p.log
