#!/bin/bash
# round 3, call T: force integration rides in the colouring rounds (guest workgroups); parity subset, A/B bench, timeline
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step_graphs.py tests/test_gpu_sharding.py tests/test_capi_symbols.py -q -m gpu -x > gpurun_out/r3t_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3t_pytest.log
tail -3 gpurun_out/r3t_pytest.log
for v in 1 0 1 0; do MI_FORCES_GUEST=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest 2>/dev/null | tail -1 > gpurun_out/r3t_bench_$v.json; python - <<PY
import json; d=json.load(open("gpurun_out/r3t_bench_$v.json")); print("guest=$v", round(d["value"],1), round(d["ms_per_step"],4), {k: round(x,4) for k,x in d["stage_ms"].items()})
PY
done
bash tools/gpu_timeline.sh > /dev/null 2>&1; cut -c1-100 gpurun_out/timeline.txt
