#!/bin/bash
# dev experiment: instruction / wait counters of k_bp_pairs_grid at the 240-step and the at-rest state
ulimit -c 0; mkdir -p gpurun_out; export TMPDIR=/tmp
for SET in 240 1500; do
 for CTR in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_WR"; do
  RAW=/tmp/pmc_bp; rm -rf $RAW
  timeout 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $RAW -o x -- python bench.py --steps 4 --warmup 2 --settle $SET --no-cpu-baseline --no-at-rest > /dev/null 2>&1
  python - "$SET" <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/pmc_bp/**/x_counter_collection.csv", recursive=True)
if not f: print("no output"); sys.exit()
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0]
    if "pairs_grid" in k or "narrow_clip" in k: vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in vals.items():
    print(sys.argv[1], k, {c: round(sum(v[-5:]) / len(v[-5:])) for c, v in d.items()})
PY
 done
done
