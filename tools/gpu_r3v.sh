#!/bin/bash
# round 3, call V: soaks — exact seam (400 steps, 3 scenes), then the general soak (pile 3000 steps; small scenes graph vs plain)
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python tools/exp_exact_soak.py 400 > gpurun_out/r3v_exact_soak.log 2>&1; tail -4 gpurun_out/r3v_exact_soak.log | cut -c1-500
PILE_STEPS=3000 SMALL_STEPS=2000 WITH_TORCH=1 timeout 900 python tools/gpu_soak.py > gpurun_out/r3v_soak.log 2>&1; tail -2 gpurun_out/r3v_soak.log | cut -c1-1500
