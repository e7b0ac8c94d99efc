#!/bin/bash
# dev helper: GPU tests + bench on the GPU box
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
python bench.py --steps 30 --warmup 250 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_full.log
