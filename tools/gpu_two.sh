#!/bin/bash
# dev helper: a subset of the GPU tests
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "xcd_partitioned" 2>&1 | grep -E "^E|passed|failed" | head -12
