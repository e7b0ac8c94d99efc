#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/ko.py <<'PY'
import sys, time, os
sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
tag = sys.argv[1]
sc = scenes.obb_pile(128, 16, 128)
w = sc.populate(mi.create_world(0)); s = sc.settings()
if sys.argv[2] == "save":
    w.step_fixed(s, sc.dt, 280)
    open("/tmp/ck.bin", "wb").write(w.save_checkpoint())
    print("saved", w.counts()["num_contacts"], flush=True)
else:
    w.load_checkpoint(open("/tmp/ck.bin", "rb").read())
    out = []
    for _ in range(4):
        try:
            w.step_fixed(s, sc.dt, 1); st = w.stage_times(); out.append((round(st["solve"], 4), w.counts()["num_contacts"], w.solver_kernel()))
        except Exception as e:
            out.append(str(e)[:80]); break
    print(tag, out, flush=True)
PY
run() { timeout 200 python /tmp/ko.py "$@" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3; }
run base save
run xcd load
MI_DBG_TIMELINE_OUT=gpurun_out/timeline.bin MI_PHYSICS_LIB=build_exp/libmi_physics_tl.so run xcd-tl load
python - <<'PY'
import numpy as np
a = np.fromfile("gpurun_out/timeline.bin", dtype=np.uint64).reshape(-1, 256, 8)
print("waves", a.shape[0])
valid = a[:, :, 0] != 0
print("visits per wave", valid.sum(1).min(), valid.sum(1).max())
t = a[:, :, :6].astype(np.int64)
names = ["top->rows landed", "rows->bodies issued", "issue->first check", "first check->tags ok", "tags ok->before stores"]
for i, nme in enumerate(names):
    d = (t[:, :, i + 1] - t[:, :, i])[valid] * 10.0   # ns (100 MHz)
    print(f"{nme:28s} mean {d.mean():8.1f} ns  median {np.median(d):8.1f}  p90 {np.percentile(d, 90):8.1f}")
# visit to visit
vt = t[:, :, 0]
nv = valid.sum(1)
d = []
for w in range(a.shape[0]):
    x = vt[w, :nv[w]]
    d.append(np.diff(x) * 10.0)
d = np.concatenate(d)
print("visit period mean %.1f ns median %.1f" % (d.mean(), np.median(d)))
st = (t[:, :, 5])[valid]; nxt = np.concatenate([vt[w, 1:nv[w]] - t[w, :nv[w] - 1, 5] for w in range(a.shape[0])]) * 10.0
print("before stores -> next top: mean %.1f ns median %.1f" % (nxt.mean(), np.median(nxt)))
polled = ((t[:, :, 4] - t[:, :, 3])[valid] * 10.0) > 300
print("fraction of visits that polled >300ns:", polled.mean())
span = (t[:, :, 5][valid].max() - t[:, :, 0][valid].min()) * 10.0
print("kernel span ns", span)
PY
