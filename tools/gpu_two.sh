#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
timeout 600 python -m pytest tests/test_learning.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/two.log
bash tools/gpu_learning.sh
bash tools/gpu_cfgs.sh > /dev/null 2>&1
