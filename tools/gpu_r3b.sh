#!/bin/bash
# round 3, call B: the direct-against-the-reference GPU tests (replay of the reference's order; teacher-forced one-step comparison)
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_reference_direct.py -q -m gpu -s --junitxml=gpurun_out/r3b_reference_direct.xml 2>&1 | tail -60 | tee gpurun_out/r3b_reference_direct.log
