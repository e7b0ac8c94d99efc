#!/bin/bash
# round 3, call C: the whole GPU suite (new: direct-against-reference replay / teacher-forced, compiled binding, sharding additions)
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -45 | tee gpurun_out/r3c_pytest_gpu.log
./oracle/_ref/binding_check 2>&1 | tail -8 | tee gpurun_out/r3c_binding_check.log
