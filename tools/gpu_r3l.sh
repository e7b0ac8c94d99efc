#!/bin/bash
# round 3, call L: constraint rows stored non-temporally by k_contact_init (A/B against a build with plain stores), bench state and at rest
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], "steps/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "solver us", round(d["roofline"]["avg_launch_us"], 1), "at_rest", round(d["at_rest"]["value"], 1), "rest solver us", round(d["at_rest"]["solver_avg_launch_us"], 1), "stage", {k: round(v, 3) for k, v in d["stage_ms"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3l_stream.json; show "streaming rows  " gpurun_out/r3l_stream.json
MI_PHYSICS_LIB=build_exp/libmi_nostream.so timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3l_plain.json; show "plain stores    " gpurun_out/r3l_plain.json
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3l_stream2.json; show "streaming again " gpurun_out/r3l_stream2.json
MI_PHYSICS_LIB=build_exp/libmi_nostream.so timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3l_plain2.json; show "plain again     " gpurun_out/r3l_plain2.json
bash tools/gpu_timeline.sh 2>&1 | tail -1; grep "contact_init\|solve_persist" gpurun_out/timeline.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -3
