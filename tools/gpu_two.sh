#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_learning.py -x -q -m gpu -k "trajectory or learning or deletion" 2>&1 | tail -3
bash tools/gpu_cfgs.sh > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/cfgs.json"))
for k,v in d.items(): print(k, round(v["steps_per_s"]), v["stage_ms"]["solve"])
PY
