#!/bin/bash
# one rocprofv3 --kernel-trace --stats pass of the driver-flag bench; prints the steady-state (last 20 dispatches) mean duration per kernel
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp
RAW=/tmp/prof_raw; rm -rf $RAW; mkdir -p $RAW/stats
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest"
$B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), 'steps/s; solver avg_launch_us', round(d['roofline']['avg_launch_us'],1))"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o r02 -- $B > gpurun_out/stats_bench.log 2>&1
python - <<'PY'
import csv, collections, glob
tr = sorted(glob.glob('/tmp/prof_raw/stats/**/*_kernel_trace.csv', recursive=True))
rows = list(csv.DictReader(open(tr[0]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list)
for r in rows: dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1][-20:])):
    n = min(20, len(v)); per_step = len(v) / max(1, len(dur["k_publish_readback"]))
    m = sum(v[-n:]) / n; tot += m * per_step
    if m * per_step > 4: print(f"{m:8.1f} us x{per_step:5.2f}  {k[:60]}")
print(f"sum of kernels per step: {tot:.1f} us")
PY
