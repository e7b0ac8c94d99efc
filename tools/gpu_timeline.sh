#!/bin/bash
# dev helper: kernel timeline of one settled step (rocprofv3 --kernel-trace), with the gaps between kernels
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp
RAW=/tmp/prof_tl; rm -rf $RAW; mkdir -p $RAW
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $RAW -o tl -- python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-at-rest ${TL_EXTRA} > gpurun_out/tl_bench.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_tl/**/tl_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step ends with k_publish_readback (k_reset_scalars' work rides in it): take a full step of the timed region, before the six extra (stage-timed, step-timed) steps
idx = [i + 1 for i, r in enumerate(rows) if "k_publish_readback" in r["Kernel_Name"]]
a, b = idx[-36], idx[-35]   # (the run ends with 3 + 3 sampled steps and 20 steps of the whole-step-event region: this one lies in the timed region)
t0 = int(rows[a]["Start_Timestamp"])
out = []
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%8.1f  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:70]))
    prev_end = e
open("gpurun_out/timeline.txt", "w").write("\n".join(out) + "\n")
print("step span us:", (prev_end - t0) / 1e3, "kernels:", b - a)
PY
