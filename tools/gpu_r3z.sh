#!/bin/bash
# round 3, call Z: joint data as one untyped block (no scratch in the island kernels) — parity subset, cfg4 / cfg5 timings
ulimit -c 0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_learning.py -q -m gpu -x -k "island or ragdoll or vehicle or joint or constraint or learning or cfg4 or cfg5 or motor or edits" > gpurun_out/r3z_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3z_pytest.log
tail -3 gpurun_out/r3z_pytest.log
CFGS=cfg4,cfg5 bash tools/gpu_cfgs.sh > gpurun_out/r3z_cfgs.log 2>&1
grep -E "cfg4|cfg5" gpurun_out/r3z_cfgs.log | cut -c1-650
