#!/bin/bash
# dev helper: GPU tests + bench + rocprofv3 kernel stats on the GPU box (one gpurun call)
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 30 --warmup 250 --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/bench_full.log
MI_SOLVER=launch timeout 300 python bench.py --steps 30 --warmup 250 --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/bench_launch.log
RAW=/tmp/prof_raw; rm -rf $RAW; mkdir -p $RAW
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o r01 -- python bench.py --steps 10 --warmup 250 --no-cpu-baseline > gpurun_out/stats_bench.log 2>&1
python tools/summarize_prof.py $RAW gpurun_out/prof_summary
