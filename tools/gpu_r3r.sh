#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
timeout 600 python tools/exp_exact.py > gpurun_out/r3r_exact.log 2>&1
tail -20 gpurun_out/r3r_exact.log | cut -c1-600
