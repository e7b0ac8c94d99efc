#!/bin/bash
# A/B of library builds (build_variants/*.so against the in-tree library): one rocprofv3 --kernel-trace pass of the driver-flag bench each;
# prints the steady-state (last 20 dispatches) mean of the kernels matching $KERNELS (regex) and the kernel sum per step
ulimit -c 0
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest"
shopt -s nullglob
for lib in "" build_variants/*.so; do
  RAW=/tmp/prof_var; rm -rf $RAW; mkdir -p $RAW
  if [ -n "$lib" ]; then export MI_PHYSICS_LIB=$PWD/$lib; else unset MI_PHYSICS_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $RAW -o v -- $B > /tmp/var.log 2>&1
  LIBNAME=${lib:-in-tree} python - <<'PY'
import csv, collections, glob, os, re
tr = sorted(glob.glob('/tmp/prof_var/**/*_kernel_trace.csv', recursive=True))
rows = list(csv.DictReader(open(tr[0]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list)
for r in rows: dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
steps = max(1, len(dur["mi::k_world_colliders"]))
tot = sum(sum(v[-min(20, len(v)):]) / min(20, len(v)) * len(v) / steps for v in dur.values())
pat = re.compile(os.environ.get("KERNELS", "contact_init"))
sel = {k.replace("mi::", "").replace("void ", "")[:28]: round(sum(v[-20:]) / min(20, len(v)), 1) for k, v in dur.items() if pat.search(k)}
print(f"{os.environ['LIBNAME']:36s} kernels/step {tot:7.1f} us  {sel}")
PY
done
