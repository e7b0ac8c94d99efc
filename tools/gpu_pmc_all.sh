#!/bin/bash
# dev experiment: per kernel, VALU instructions vs wave cycles vs waits (bench state and at rest): which kernels stall rather than work
ulimit -c 0; mkdir -p gpurun_out; export TMPDIR=/tmp
for SET in 240 1500; do
  RAW=/tmp/pmc_all; rm -rf $RAW
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $RAW -o x -- python bench.py --steps 4 --warmup 2 --settle $SET --no-cpu-baseline --no-at-rest > /dev/null 2>&1
  python - "$SET" <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/pmc_all/**/x_counter_collection.csv", recursive=True)
t = glob.glob("/tmp/pmc_all/**/x_kernel_trace.csv", recursive=True)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    vals[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(t[0])):
    dur[r["Kernel_Name"].split("(")[0][:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("state", sys.argv[1])
rows = []
for k, d in vals.items():
    m = {c: sum(v[-5:]) / len(v[-5:]) for c, v in d.items()}
    us = sum(dur[k][-5:]) / max(1, len(dur[k][-5:]))
    rows.append((us, k, m))
for us, k, m in sorted(rows, reverse=True)[:16]:
    valu, cyc, wa, wi = m.get("SQ_INSTS_VALU", 0), m.get("SQ_WAVE_CYCLES", 0), m.get("SQ_WAIT_ANY", 0), m.get("SQ_WAIT_INST_ANY", 0)
    print("%8.1f us  %-44s valu %6.1fM  wave_cyc %7.1fM  issue%% %4.1f  wait_any%% %4.1f  wait_inst%% %4.1f" % (us, k, valu / 1e6, cyc / 1e6, 100 * valu / max(cyc, 1), 100 * wa / max(cyc, 1), 100 * wi / max(cyc, 1)))
PY
done
