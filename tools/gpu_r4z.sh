#!/bin/bash
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r4z_pytest_gpu.log 2>&1; tail -12 gpurun_out/r4z_pytest_gpu.log
