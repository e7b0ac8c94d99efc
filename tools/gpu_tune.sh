#!/bin/bash
# dev helper: knock-out experiments on the dataflow solver (trace build; results are wrong for dbg != 0, timing only)
ulimit -c 0
mkdir -p gpurun_out; : > gpurun_out/tune.log
cat > /tmp/tr.py <<'PY'
import sys, os, pathlib
sys.path.insert(0, ".")
os.environ["MI_PHYSICS_LIB"] = "d3d12renderer_amd/libmi_physics_trace.so"
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.obb_pile(128, 16, 128, solver_iterations=20)
w = sc.populate(mi.create_world(0))
w.step_fixed(sc.settings(), sc.dt, 250)
dbg = os.environ.get("DBG", "0")
os.environ["MI_FLOW_DBG"] = dbg
acc = 0.0; n = 0
for i in range(6):
    try:
        w.step_fixed(sc.settings(), sc.dt, 1)
    except Exception as e:
        print("dbg", dbg, "error", str(e)[-60:]); break
    acc += w.stage_times()["solve"]; n += 1
print("dbg", dbg, "solve_ms", round(acc / max(n, 1), 3), "steps", n)
PY
for d in 0 16 48 24; do DBG=$d timeout 100 python /tmp/tr.py 2>&1 | tail -1 >> gpurun_out/tune.log; done
cat gpurun_out/tune.log
