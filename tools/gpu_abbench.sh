#!/bin/bash
# dev helper: A/B of one environment switch on the headline bench (the driver's flags), alternating runs:  AB_VAR=MI_FUSE_WORLD AB_VALS="unset 0 unset 0" bash tools/gpu_abbench.sh
ulimit -c 0; mkdir -p gpurun_out
for v in ${AB_VALS:-unset 0 unset 0}; do
  if [ "$v" = unset ]; then unset $AB_VAR; else export $AB_VAR=$v; fi
  timeout 600 python bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-cpu-baseline --no-at-rest 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$AB_VAR=$v', round(d['value'], 1), 'steps/s', round(d['ms_per_step'], 4), 'ms  solver', round(d['roofline']['avg_launch_us'], 1), {k: round(x, 3) for k, x in d['stage_ms'].items()})"
done
