#!/bin/bash
# round 3, call D: A/B of the schedule-stage changes (fused colouring of short lists, partition launch skipped, 16-byte history slots), parity after them
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-at-rest"
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1], "steps/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "dev ms", round(d["device_ms_per_step"], 4), "solver us", round(d["roofline"]["avg_launch_us"], 1), "stage", {k: round(v, 3) for k, v in d["stage_ms"].items()})
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
MI_COLOR_LIST=0 MI_SKIP_PARTITION=0 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3d_old_path.json; show "old launches " gpurun_out/r3d_old_path.json
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3d_new.json; show "new default  " gpurun_out/r3d_new.json
MI_COLOR_LIST=0 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3d_nolist.json; show "no list      " gpurun_out/r3d_nolist.json
MI_NO_TIMES=1 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3d_notimes.json; show "no times     " gpurun_out/r3d_notimes.json
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3d_new2.json; show "new default 2" gpurun_out/r3d_new2.json
bash tools/gpu_timeline.sh 2>&1 | tail -2
cp gpurun_out/timeline.txt gpurun_out/r3d_timeline.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_step_graphs.py tests/test_scene_formats.py -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/r3d_pytest.log
timeout 900 python -m pytest tests/test_gpu_reference_direct.py -q -m gpu -s -k "one_step" 2>&1 | grep "teacher-forced\|passed\|failed" | tee gpurun_out/r3d_teacher_forced.log
