#!/bin/bash
# dev helper: per-visit wall-clock stamps of the persistent solver (library built with -DMI_DBG_TIMELINE into build_exp/)
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/tl.py <<'PY'
import sys
sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.obb_pile(128, 16, 128)
w = sc.populate(mi.create_world(0)); s = sc.settings()
w.step_fixed(s, sc.dt, 285)
print("solve", w.stage_times()["solve"], w.solver_kind())
PY
MI_DBG_TIMELINE_STEP=283 MI_DBG_TIMELINE_OUT=gpurun_out/timeline.bin MI_PHYSICS_LIB=build_exp/libmi_physics_tl.so timeout 200 python /tmp/tl.py 2>&1 | tail -1
python - <<'PY'
import numpy as np
a = np.fromfile("gpurun_out/timeline.bin", dtype=np.uint64).reshape(-1, 256, 8)
valid = a[:, :, 0] != 0
t = a[:, :, :6].astype(np.int64) * 10
names = ["top->rows read (vmcnt4 + acc reads)", "rows read->body loads... (stamp order: 2 precedes 1)", "issue->first check", "first check->tags ok", "tags ok->before stores"]
for i, nme in enumerate(names):
    d = (t[:, :, i + 1] - t[:, :, i])[valid]
    print(f"{nme:50s} mean {d.mean():8.1f} ns  median {np.median(d):8.1f}  p90 {np.percentile(d, 90):8.1f}")
vt = t[:, :, 0]; nv = valid.sum(1)
d = np.concatenate([np.diff(vt[w, :nv[w]]) for w in range(a.shape[0])])
print("visit period mean %.1f ns median %.1f" % (d.mean(), np.median(d)))
nxt = np.concatenate([vt[w, 1:nv[w]] - t[w, :nv[w] - 1, 5] for w in range(a.shape[0])])
print("before stores -> next top: mean %.1f ns median %.1f" % (nxt.mean(), np.median(nxt)))
d02 = (t[:, :, 2] - t[:, :, 0])[valid]; print("top -> body loads issued: mean %.1f median %.1f" % (d02.mean(), np.median(d02)))
d23 = (t[:, :, 3] - t[:, :, 2])[valid]; print("body loads issued -> first check: mean %.1f median %.1f" % (d23.mean(), np.median(d23)))
d34 = (t[:, :, 4] - t[:, :, 3])[valid]; print("polling: mean %.1f median %.1f, fraction > 300 ns: %.2f" % (d34.mean(), np.median(d34), (d34 > 300).mean()))
print("span", (t[:, :, 5][valid].max() - t[:, :, 0][valid].min()))
PY
rm -f gpurun_out/timeline.bin gpurun_out/timeline.bin.knock   # (16 MB each: gpurun_out/ travels back only while it stays under 64 MiB)
