#!/bin/bash
# round 3, call N: full GPU suite on HEAD + what a rank's step costs as the replicated scene grows (one GPU plays the middle tile)
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r3n_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3n_pytest.log
tail -5 gpurun_out/r3n_pytest.log
timeout 600 python tools/exp_weak.py > gpurun_out/r3n_weak.log 2>&1; cp gpurun_out/exp_weak.json gpurun_out/r3n_weak.json
cat gpurun_out/r3n_weak.log | cut -c1-900
