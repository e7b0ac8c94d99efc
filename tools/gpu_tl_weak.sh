#!/bin/bash
# dev helper: kernel timeline of one step of the middle tile of an R-tile replicated pile (tools/exp_weak.py R)
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp
export R=${1:-8}
RAW=/tmp/prof_tlw; rm -rf $RAW; mkdir -p $RAW
WEAK_RANK=${WEAK_RANK:-} timeout 600 rocprofv3 --kernel-trace --output-format csv -d $RAW -o tl -- python tools/exp_weak.py $R > gpurun_out/tlw.log 2>&1
tail -2 gpurun_out/tlw.log
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_tlw/**/tl_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i + 1 for i, r in enumerate(rows) if "k_publish_readback" in r["Kernel_Name"]]
a, b = idx[-5], idx[-4]
t0 = int(rows[a]["Start_Timestamp"])
out = []
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%8.1f  +%6.1f gap  %7.1f us  %s  grid %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:60], r.get("Grid_Size", "")))
    prev_end = e
open("gpurun_out/timeline_weak_%s.txt" % __import__("os").environ.get("R", "8"), "w").write("\n".join(out) + "\n")
print("step span us:", (prev_end - t0) / 1e3, "kernels:", b - a)
PY
