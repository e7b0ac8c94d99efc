#!/bin/bash
# round 3, call J: LDS-staged pair pass, deferred step timing; parity
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
timeout 300 $B 2>gpurun_out/r3j_err.log | tail -1 > gpurun_out/r3j_new.json; show "HEAD            " gpurun_out/r3j_new.json
MI_BP_LDS=0 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3j_nolds.json; show "pair pass via L2" gpurun_out/r3j_nolds.json
MI_EAGER_TIMES=1 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3j_eager.json; show "eager times     " gpurun_out/r3j_eager.json
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3j_new2.json; show "HEAD again      " gpurun_out/r3j_new2.json
bash tools/gpu_timeline.sh 2>&1 | tail -2
cp gpurun_out/timeline.txt gpurun_out/r3j_timeline.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sharding.py tests/test_gpu_step_graphs.py tests/test_gpu_reference_direct.py -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/r3j_pytest.log
