#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "trajectory or golden or ray or heightmap" 2>&1 | tail -5 | tee gpurun_out/two.log
bash tools/gpu_cfgs.sh 2>&1 | grep "cfg4\|cfg5\|terrain" | cut -c1-90
python - <<'PY'
import json
d=json.load(open("gpurun_out/cfgs.json"))
for k,v in d.items(): print(k, round(v["steps_per_s"]), v["stage_ms"]["narrowphase"])
PY
