#!/bin/bash
# scratch: run selected GPU tests
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
MI_DEBUG_SYNC=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "terrain or heightmap or trajectory" 2>&1 | tail -30 | tee gpurun_out/two.log
