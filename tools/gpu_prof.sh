#!/bin/bash
# dev helper: tests + bench + rocprofv3 kernel stats of the bench command
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
python bench.py --steps 30 --warmup 250 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_full.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python bench.py --steps 10 --warmup 250 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
ls -R gpurun_out/prof | head -30
