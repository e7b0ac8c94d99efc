#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharding.py -q -m gpu -x -k "exact or misuse or rejects" > gpurun_out/r3x_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3x_pytest.log
tail -3 gpurun_out/r3x_pytest.log
timeout 900 python tools/exp_exact_cost.py > gpurun_out/r3x_cost.log 2>&1
cut -c1-300 gpurun_out/r3x_cost.log | tail -8
