#!/bin/bash
# round 3, call S: exact seam — GPU tests, then what it costs
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharding.py -q -m gpu -x > gpurun_out/r3s_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3s_pytest.log
tail -4 gpurun_out/r3s_pytest.log
timeout 900 python tools/exp_exact_cost.py > gpurun_out/r3s_cost.log 2>&1
cut -c1-600 gpurun_out/r3s_cost.log | tail -12
