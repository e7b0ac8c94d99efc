#!/bin/bash
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python -m pytest tests/test_gpu_sharding.py -q -m gpu -k "loops_back" > gpurun_out/r4x_loopback.log 2>&1; tail -15 gpurun_out/r4x_loopback.log
for mg in 3 2 1 0; do
MI_COLOR_MARGIN=$mg timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-at-rest 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('margin $mg', round(d['value'],1), 'steps/s', 'reruns', d['step_modes_timed']['synchronous_reruns'], 'sched us', round(d['stage_ms']['schedule']*1e3,1), 'median ms', round(d['step_ms_median'],4))" | tee -a gpurun_out/r4x_margin.log
done
