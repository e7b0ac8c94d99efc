#!/bin/bash
# dev helper: do the steps of a scene replay as HIP graphs?  (MI_GRAPH_DEBUG prints where consecutive signatures differ and the totals)
export TMPDIR=/tmp
cat > /tmp/graph_dbg.py <<'PY'
import sys, os, time
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
name = os.environ.get("SCENE", "sphere_drop"); args = [int(x) for x in os.environ.get("SCENE_ARGS", "16").split(",") if x]
sc = getattr(scenes, name)(*args)
w = sc.populate(mi.create_world(0)); s = sc.settings()
warm = int(os.environ.get("WARM", "240")); steps = int(os.environ.get("STEPS", "200"))
w.step_fixed(s, sc.dt, warm); w.counts()
t0 = time.perf_counter(); w.step_fixed(s, sc.dt, steps); w.counts(); dt = (time.perf_counter() - t0) / steps
print(name, args, "ms/step", round(dt * 1e3, 4), w.counts(), "kind", w.solver_kind(), w.stage_times())
w.close()
PY
python /tmp/graph_dbg.py 2>&1 | tail -${TAIL:-25}
