#!/bin/bash
# dev helper: per-kernel PMC counters (one pass per counter) of the middle tile of an R-tile replicated pile against the single world (tools/exp_weak.py R)
ulimit -c 0; mkdir -p gpurun_out; export TMPDIR=/tmp
for R in ${RS:-1 8}; do for C in ${COUNTERS:-FETCH_SIZE WRITE_SIZE}; do
  RAW=/tmp/pmcw_${R}_$C; rm -rf $RAW; mkdir -p $RAW
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $RAW -o p -- env WEAK_RANK=${WEAK_RANK:-} python tools/exp_weak.py $R > /dev/null 2>&1
  python - "$RAW" "$R" "$C" <<'PY'
import csv, glob, sys, collections
raw, R, C = sys.argv[1:]
f = glob.glob(raw + "/**/p_counter_collection.csv", recursive=True)[0]
v = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") == C: v[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
out = {k.replace("void ", "").replace("mi::", "")[:34]: round(sum(x[-20:]) / len(x[-20:]), 1) for k, x in v.items() if len(x) > 100}
print(R, C, out)
PY
done; done
