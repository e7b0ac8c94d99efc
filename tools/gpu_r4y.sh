#!/bin/bash
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python -m pytest tests/test_gpu_sharding.py -q -m gpu > gpurun_out/r4y_sharding.log 2>&1; tail -12 gpurun_out/r4y_sharding.log
