#!/bin/bash
# dev helper: one traced solver launch (per-wave timestamps) at steady state
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/tr.py <<'PY'
import sys, os, pathlib
sys.path.insert(0, ".")
import d3d12renderer_amd as mi
mi.LIB_PATH = pathlib.Path(os.environ.get("TRACE_LIB", "d3d12renderer_amd/libmi_physics_trace.so")).resolve()
from d3d12renderer_amd import scenes
sc = scenes.obb_pile(128, 16, 128, solver_iterations=20)
w = sc.populate(mi.create_world(0))
w.step_fixed(sc.settings(), sc.dt, 262)
print(w.counts(), w.stage_times())
PY
MI_FLOW_TRACE_FILE=/tmp/trace.bin MI_FLOW_TRACE_STEP=258 timeout 300 python /tmp/tr.py 2>&1 | tail -2
python - <<'PY'
import numpy as np, struct
b = open("/tmp/trace.bin", "rb").read()
T, P, W, _ = struct.unpack("4I", b[:16])
desc = np.frombuffer(b[16:16 + 8 * T], np.uint32).reshape(T, 2)
t = np.frombuffer(b[16 + 8 * T:], np.uint64).reshape(P, T, 8).astype(np.int64)
t0 = t[..., 0].min()
t = (t - t0) * 0.01   # us (100 MHz)
print("tiles", T, "sweeps", P, "launch span us", t[..., 5].max())
life = t[..., 5] - t[..., 0]
print("wave lifetime us: mean %.2f p50 %.2f p95 %.2f" % (life.mean(), np.median(life), np.percentile(life, 95)))
for a, b_, name in ((0, 1, "start->meta landed"), (1, 2, "imp+bodies round trip (lands rows)"), (2, 3, "dependency wait"), (3, 4, "compute"), (4, 5, "publish (issue)")):
    d = t[..., b_] - t[..., a]
    print("%-32s mean %.2f p50 %.2f p95 %.2f max %.2f" % (name, d.mean(), np.median(d), np.percentile(d, 95), d.max()))
# per sweep: start of first tile, end of last tile
for s in (0, 1, 2, 10, 19):
    print("sweep", s, "first start %.1f last start %.1f first end %.1f last end %.1f" % (t[s, :, 0].min(), t[s, :, 0].max(), t[s, :, 5].min(), t[s, :, 5].max()))
# residency: number of waves alive over time
ev = np.concatenate([np.stack([t[..., 0].ravel(), np.ones(t[..., 0].size)], 1), np.stack([t[..., 5].ravel(), -np.ones(t[..., 0].size)], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1])
print("resident waves: mean %.0f max %.0f" % (alive.mean(), alive.max()))
# dependency wait by colour position within sweep 10
s = 10
dw = t[s, :, 3] - t[s, :, 2]
chunks = np.array_split(np.arange(T), 12)
print("dep wait by tile-index dodecile (sweep 10):", [round(float(dw[c].mean()), 2) for c in chunks])
st = t[s, :, 0]
print("start time by dodecile:", [round(float(st[c].mean()), 1) for c in chunks])
np.save("gpurun_out/trace_sweep10.npy", t[10].astype(np.float32))
PY
