#!/bin/bash
# round 3, call P: sharded-world early-outs (classify / prepare / forces), owner counts inside k_manifold_keys, headers + axis in one launch
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r3p_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3p_pytest.log
tail -4 gpurun_out/r3p_pytest.log
timeout 600 python tools/exp_weak.py > gpurun_out/r3p_weak.log 2>&1; cp gpurun_out/exp_weak.json gpurun_out/r3p_weak.json
cut -c1-700 gpurun_out/r3p_weak.log
bash tools/gpu_tl_weak.sh 8
