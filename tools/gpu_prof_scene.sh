#!/bin/bash
# dev helper: rocprofv3 kernel stats of a named scene (SCENE=terrain_big STEPS=60 WARM=300)
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/scene.py <<'PY'
import sys, os
sys.path.insert(0, ".")
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = getattr(scenes, os.environ.get("SCENE", "terrain_big"))()
w = sc.populate(mi.create_world(0)); s = sc.settings()
w.step_fixed(s, sc.dt, int(os.environ.get("WARM", "300")) + int(os.environ.get("STEPS", "60")))
print(w.counts(), w.stage_times())
PY
RAW=/tmp/prof_scene; rm -rf $RAW; mkdir -p $RAW
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o s -- python /tmp/scene.py > gpurun_out/scene_prof.log 2>&1
F=$(find $RAW -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY' | tee gpurun_out/scene_kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):7d} avg_us {float(r["AverageNs"])/1e3:9.2f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} pct {float(r["Percentage"]):5.1f}')
PY
