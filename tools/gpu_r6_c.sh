#!/bin/bash
# round 6: A/B of an environment switch at the driver's flags (AB_VAR name, values 0/1 alternating), after a parity subset
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
T0=$(date +%s)
if [ -n "$AB_TESTS" ]; then timeout 1200 python -m pytest $AB_TESTS -q -m gpu -x 2>&1 | tail -4; echo "tests at $(( $(date +%s) - T0 )) s"; fi
for i in 1 2 3; do for V in 0 1; do
env $AB_VAR=$V timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${AB_EXTRA:---no-at-rest} 2>/dev/null | tail -1 > gpurun_out/ab_$V.json
python - <<PY
import json
d = json.load(open("gpurun_out/ab_$V.json"))
print("$AB_VAR=$V", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "solver us", round(d["roofline"]["avg_launch_us"], 1), "dev", round(d["device_ms_per_step"], 4), "at_rest", (d.get("at_rest") or {}).get("value"), (d.get("at_rest") or {}).get("solver_avg_launch_us"))
PY
done; done
echo "done at $(( $(date +%s) - T0 )) s"
