#!/bin/bash
# development: kernel times of one fuzz world (tools/exp/fuzz_time.py SEED) by rocprofv3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_f; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o f -- python $GRAFT_REPO_ROOT/tools/exp/fuzz_time.py $1 > /tmp/f.log 2>&1
python - <<PY
import csv,glob
fs=glob.glob("/tmp/prof_f/**/*kernel_stats.csv",recursive=True)
print(fs)
rows=list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:8]: print(r["Name"][:70], r["Calls"], "total ms", round(float(r["TotalDurationNs"])/1e6,2), "max ms", round(float(r["MaxNs"])/1e6,2))
PY
