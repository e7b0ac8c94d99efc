#!/bin/bash
# dev helper: A/B of one environment switch on the other configs:  AB_VAR=MI_GJK_WAVE AB_VALS="unset 0 1" CFGS=cfg4,cfg5 bash tools/gpu_ab.sh
ulimit -c 0; mkdir -p gpurun_out
for v in ${AB_VALS:-unset 0}; do
  if [ "$v" = unset ]; then unset $AB_VAR; else export $AB_VAR=$v; fi
  echo "== $AB_VAR=$v"; bash tools/gpu_cfgs.sh 2>&1 | grep -E "^(cfg|terrain|zones|pile)" | python -c "
import sys, json
for l in sys.stdin:
    n, j = l.split(' ', 1); d = json.loads(j); print('  ', n, round(d['ms_per_step'], 4), 'ms', {k: round(v, 3) for k, v in d['stage_ms'].items()})"
done
