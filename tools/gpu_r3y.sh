#!/bin/bash
# round 3, call Y: private joint islands — parity (small + full size + learning), then the other configs' timings
ulimit -c 0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_learning.py tests/test_gpu_step_graphs.py -q -m gpu -x > gpurun_out/r3y_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3y_pytest.log
tail -4 gpurun_out/r3y_pytest.log
bash tools/gpu_cfgs.sh > gpurun_out/r3y_cfgs.log 2>&1; cp gpurun_out/cfgs.json gpurun_out/r3y_other_configs.json 2>/dev/null
grep -E "cfg4|cfg5|cfg1|zones" gpurun_out/r3y_cfgs.log | cut -c1-700
