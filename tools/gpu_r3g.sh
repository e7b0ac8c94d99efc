#!/bin/bash
# round 3, call G: fused schedule tail (k_schedule_finish), table clear inside k_narrow, pair statistics beside k_emit_manifolds; colour-round margin; parity
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
B="python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-at-rest"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], "steps/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "dev ms", round(d["device_ms_per_step"], 4), "solver us", round(d["roofline"]["avg_launch_us"], 1), "reruns", d["step_modes_timed"]["synchronous_reruns"], "stage", {k: round(v, 3) for k, v in d["stage_ms"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
timeout 300 $B 2>gpurun_out/r3g_err.log | tail -1 > gpurun_out/r3g_new.json; show "HEAD           " gpurun_out/r3g_new.json
MI_COLOR_MARGIN=1 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3g_margin1.json; show "colour margin 1" gpurun_out/r3g_margin1.json
MI_COLOR_MARGIN=2 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3g_margin2.json; show "colour margin 2" gpurun_out/r3g_margin2.json
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3g_new2.json; show "HEAD again     " gpurun_out/r3g_new2.json
bash tools/gpu_timeline.sh 2>&1 | tail -2
cp gpurun_out/timeline.txt gpurun_out/r3g_timeline.txt
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r3g_pytest.log
