#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent" 2>&1 | tail -5 | tee gpurun_out/two.log
