#!/bin/bash
# dev helper: where a batched learning step (4096 ragdoll environments) spends its time: kernel totals vs wall clock
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/learn_prof.py <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np
from d3d12renderer_amd.learning import PhysicsDLL
d = PhysicsDLL(); d.seed(1)
_, _, amin, amax = d.ranges()
n = 4096
d.reset_batch(n)
rng = np.random.default_rng(0)
acts = [(rng.uniform(-1, 1, (n, 27)) * 0.15 * (amax - amin)).astype(np.float32) for _ in range(8)]
for i in range(30): d.step_batch(acts[i % 8])
t0 = time.perf_counter()
for i in range(50): d.step_batch(acts[i % 8])
print("wall ms per env step", (time.perf_counter() - t0) / 50 * 1e3)
PY
RAW=/tmp/prof_learn; rm -rf $RAW; mkdir -p $RAW
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o s -- python /tmp/learn_prof.py 2>&1 | grep "wall ms"
F=$(find $RAW -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per env step (80 steps):", tot / 80 / 1e6, "ms")
for r in rows[:12]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):6d} avg_us {float(r["AverageNs"])/1e3:9.2f} per-step_ms {float(r["TotalDurationNs"])/80/1e6:7.3f}')
PY
