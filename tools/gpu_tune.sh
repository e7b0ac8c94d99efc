#!/bin/bash
# dev helper: latency probe — small scenes, both solver paths
ulimit -c 0
mkdir -p gpurun_out; : > gpurun_out/tune.log
for g in "16 16 16" "32 16 32" "64 16 64"; do
 for m in flow launch; do
  echo "grid $g solver $m" >> gpurun_out/tune.log
  MI_SOLVER=$m MI_FLOW_TUNE=0 MI_FLOW_LDS=54000 timeout 200 python bench.py --grid $g --steps 20 --warmup 245 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'solve', round(d['stage_ms']['solve'],3), 'colors', d['config']['colors'], 'manifolds', d['config']['manifolds'], 'launches', d['roofline']['launches_per_step'], {k:round(v,3) for k,v in d['stage_ms'].items()})" >> gpurun_out/tune.log 2>&1
 done
done
cat gpurun_out/tune.log
