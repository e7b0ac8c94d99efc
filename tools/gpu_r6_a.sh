#!/bin/bash
# round 6, call A: the new sharding test, a baseline bench on this box, the knock-out harness
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_sharding.py -q -m gpu -k "sweep_axis or block_skipping" 2>&1 | tail -3
echo "tests at $(( $(date +%s) - T0 )) s"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest 2>/dev/null | tail -1 > gpurun_out/r6a_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6a_bench.json"))
print("bench", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "solver us", round(d["roofline"]["avg_launch_us"], 1), "bound", d["roofline"]["bound"], "chain", d["roofline"]["chain"], "l2", d.get("with_whole_step_events"))
print("stages", {k: round(v, 4) for k, v in d["stage_ms"].items()})
PY
echo "bench at $(( $(date +%s) - T0 )) s"
bash tools/gpu_knockout.sh 2>&1 | tail -60
echo "done at $(( $(date +%s) - T0 )) s"
