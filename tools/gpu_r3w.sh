#!/bin/bash
# round 3, call W: exact seam through per-sweep launches of the persistent kernel — tests, soak, cost
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r3w_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3w_pytest.log
tail -3 gpurun_out/r3w_pytest.log
timeout 600 python tools/exp_exact_soak.py 200 > gpurun_out/r3w_exact_soak.log 2>&1; tail -3 gpurun_out/r3w_exact_soak.log | cut -c1-300
timeout 900 python tools/exp_exact_cost.py > gpurun_out/r3w_cost.log 2>&1
cut -c1-420 gpurun_out/r3w_cost.log | tail -8
