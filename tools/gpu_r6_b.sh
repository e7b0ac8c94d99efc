#!/bin/bash
# round 6: full GPU suite + the driver-flag bench + stage times (B_EXTRA: extra commands)
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
T0=$(date +%s)
timeout 1700 python -m pytest tests -q -m gpu -x --durations=6 > gpurun_out/r6_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_pytest.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/r6_pytest.log | tail -8
echo "suite at $(( $(date +%s) - T0 )) s"
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest 2>/dev/null | tail -1 > gpurun_out/r6_bench_$i.json
python - <<PY
import json
d = json.load(open("gpurun_out/r6_bench_$i.json"))
print("bench", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "solver us", round(d["roofline"]["avg_launch_us"], 1), "dev", round(d["device_ms_per_step"], 4))
print("stages", {k: round(v, 4) for k, v in d["stage_ms"].items()})
PY
done
echo "done at $(( $(date +%s) - T0 )) s"
