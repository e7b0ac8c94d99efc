#!/bin/bash
# round 3, call F: clip-kernel occupancy variants (MI_PHYSICS_LIB), the other configs at full size, terrain with cylinders / hulls parity
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
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3f_base.json; show "clip occ 3 (HEAD)" gpurun_out/r3f_base.json
MI_PHYSICS_LIB=build_exp/libmi_clip_w4.so timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3f_w4.json; show "clip occ 4       " gpurun_out/r3f_w4.json
MI_PHYSICS_LIB=build_exp/libmi_clip_w5.so timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3f_w5.json; show "clip occ 5       " gpurun_out/r3f_w5.json
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/r3f_base2.json; show "clip occ 3 again " gpurun_out/r3f_base2.json
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "terrain or heightmap or parity" 2>&1 | tail -4 | tee gpurun_out/r3f_pytest.log
bash tools/gpu_cfgs.sh 2>&1 | tail -10
cp gpurun_out/cfgs.json gpurun_out/r3f_other_configs.json
