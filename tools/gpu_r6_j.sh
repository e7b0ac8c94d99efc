#!/bin/bash
# round 6, last minutes: the fuzzers at HEAD on fresh seed ranges (sharded + seam worlds, 20 x worlds, small worlds with 100 steps)
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
timeout 130 python tools/gpu_fuzz_sharded.py --seeds 12000:13000 --budget 100 --out gpurun_out/j_fuzz_sharded.json 2>&1 | tail -1 | cut -c1-300
timeout 130 python tools/gpu_fuzz.py --seeds 5000:5400 --scale 20 --steps 25 --budget 100 --out gpurun_out/j_fuzz_large.json 2>&1 | tail -1 | cut -c1-300
timeout 130 python tools/gpu_fuzz.py --seeds 50000:52000 --steps 100 --budget 100 --out gpurun_out/j_fuzz_small_100_steps.json 2>&1 | tail -1 | cut -c1-300
echo "all done at $(( $(date +%s) - T0 )) s"
