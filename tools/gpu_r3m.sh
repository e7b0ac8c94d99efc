#!/bin/bash
# round 3, call M: k_integrate_forces as a guest of k_emit_manifolds (A/B by MI_FORCES_IN_EMIT=0), full GPU suite
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], "steps/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "solver us", round(d["roofline"]["avg_launch_us"], 1), "at_rest", round(d["at_rest"]["value"], 1), "stage", {k: round(v, 3) for k, v in d["stage_ms"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
timeout 300 $B 2>gpurun_out/r3m_err.log | tail -1 > gpurun_out/r3m_new.json; show "forces in emit  " gpurun_out/r3m_new.json
MI_FORCES_IN_EMIT=0 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3m_sep.json; show "forces separate " gpurun_out/r3m_sep.json
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3m_new2.json; show "forces in emit 2" gpurun_out/r3m_new2.json
MI_FORCES_IN_EMIT=0 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3m_sep2.json; show "forces separate2" gpurun_out/r3m_sep2.json
bash tools/gpu_timeline.sh 2>&1 | tail -1; cp gpurun_out/timeline.txt gpurun_out/r3m_timeline.txt; grep "emit\|integrate_forces\|manifold_keys" gpurun_out/timeline.txt
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r3m_pytest.log
