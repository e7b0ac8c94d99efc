#!/bin/bash
# dev helper: GPU tests + the weak-scaling cost experiment
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 900 python tools/exp_weak.py ${WEAK_R:-1 2 4 8} 2>&1 | tail -8 | tee gpurun_out/exp_weak.log
