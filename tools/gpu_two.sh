#!/bin/bash
# scratch: run selected GPU tests + config timings
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_learning.py tests/test_scene_formats.py -x -q -m gpu -k "trajectory or golden or learning or checkpoint or ray" > gpurun_out/two_full.log 2>&1; grep -n "Fatal\|Segmentation\|Aborted\|Memory access\|HSA\|:0:\|passed\|failed\|test_" gpurun_out/two_full.log | head -40 | tee gpurun_out/two.log
bash tools/gpu_cfgs.sh 2>&1 | grep "cfg4\|cfg5" | cut -c1-700
