#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
MI_BLOCK_DEBUG=1 timeout 300 python tools/gpu_r4aa.py > gpurun_out/r4aa.log 2>&1; grep "mi_physics" gpurun_out/r4aa.log | head -8 | cut -c1-500; tail -1 gpurun_out/r4aa.log
