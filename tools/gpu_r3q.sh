#!/bin/bash
# round 3, call Q: statistics guest as workgroup 0 of k_emit_manifolds; sharded tests, default bench twice, weak cost at 1 and 8 tiles
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_parity.py tests/test_gpu_step_graphs.py -q -m gpu -x > gpurun_out/r3q_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3q_pytest.log
tail -3 gpurun_out/r3q_pytest.log
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3q_bench$i.json; python - <<PY
import json; d=json.load(open("gpurun_out/r3q_bench$i.json")); print(d["value"], d["ms_per_step"], d.get("stage_ms"), d["roofline"]["frac"])
PY
done
timeout 600 python tools/exp_weak.py 1 8 > gpurun_out/r3q_weak.log 2>&1; cp gpurun_out/exp_weak.json gpurun_out/r3q_weak.json
cut -c1-120 gpurun_out/r3q_weak.log
bash tools/gpu_timeline.sh > /dev/null 2>&1; cat gpurun_out/timeline.txt 2>/dev/null | cut -c1-100
