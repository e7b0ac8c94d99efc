#!/bin/bash
# round 3, call H: world colliders fused into k_bp_prepare, exact colour batch; margins; parity
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-at-rest"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], "steps/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "dev ms", round(d["device_ms_per_step"], 4), "solver us", round(d["roofline"]["avg_launch_us"], 1), "reruns", d["step_modes_timed"]["synchronous_reruns"], "stage", {k: round(v, 3) for k, v in d["stage_ms"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
timeout 300 $B 2>gpurun_out/r3h_err.log | tail -1 > gpurun_out/r3h_new.json; show "HEAD            " gpurun_out/r3h_new.json
MI_FUSE_WORLD=0 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3h_nofuse.json; show "world separate  " gpurun_out/r3h_nofuse.json
MI_COLOR_MARGIN=2 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3h_margin2.json; show "colour margin 2 " gpurun_out/r3h_margin2.json
MI_COLOR_MARGIN=1 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3h_margin1.json; show "colour margin 1 " gpurun_out/r3h_margin1.json
MI_COLOR_MARGIN=1 timeout 600 python bench.py --steps 240 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3h_margin1_default.json; show "margin 1, default protocol" gpurun_out/r3h_margin1_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3h_margin1_default.json")); print("  at_rest", d.get("at_rest", {}).get("value"), "reruns", d["step_modes_timed"])
PY
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3h_new2.json; show "HEAD again      " gpurun_out/r3h_new2.json
bash tools/gpu_timeline.sh 2>&1 | tail -2
cp gpurun_out/timeline.txt gpurun_out/r3h_timeline.txt
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r3h_pytest.log
