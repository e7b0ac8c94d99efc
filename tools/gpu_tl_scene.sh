#!/bin/bash
# dev helper: kernel timeline of one settled step of a named scene (SCENE=vehicles WARM=240)
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/scene_tl.py <<'PY'
import sys, os
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = getattr(scenes, os.environ.get("SCENE", "vehicles"))()
w = sc.populate(mi.create_world(0)); s = sc.settings()
w.step_fixed(s, sc.dt, int(os.environ.get("WARM", "240")))
for _ in range(8): w.step_fixed(s, sc.dt, 1)
print(w.counts(), w.stage_times())
PY
RAW=/tmp/prof_tls; rm -rf $RAW; mkdir -p $RAW
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $RAW -o tl -- python /tmp/scene_tl.py > gpurun_out/scene_tl.log 2>&1
tail -1 gpurun_out/scene_tl.log | cut -c1-400
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_tls/**/tl_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i + 1 for i, r in enumerate(rows) if "k_publish_readback" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0; out = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%8.1f  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:70]))
    prev_end = e
open("gpurun_out/timeline_scene.txt", "w").write("\n".join(out) + "\n")
print("step span us:", (prev_end - t0) / 1e3, "kernels:", b - a)
PY
