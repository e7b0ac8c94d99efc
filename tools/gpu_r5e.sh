#!/bin/bash
# dev helper (round 5, launch diet): the new variant test + the solver variants, A/B of each new switch on the driver-flag bench, one step's timeline
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
(cd oracle && make >/dev/null 2>&1)
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "launch_variants or other_contact_solvers or retry or pose_rows" 2>&1 | tail -8 > gpurun_out/r5e_tests.txt
cat gpurun_out/r5e_tests.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest"
one() { timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'steps/s solver', round(d['roofline']['avg_launch_us'],1), {k[:5]:round(v,3) for k,v in d['stage_ms'].items()}, d['step_modes_timed']['synchronous_reruns'])"; }
{
one default
MI_FINISH_IN_NARROW=0 one nofinish
MI_COLOR_TAIL=0 one notail
MI_COLOR_TAIL_MARGIN=0 one tailmargin0
one default
MI_FINISH_IN_NARROW=0 MI_COLOR_TAIL=0 one alloff
MI_COLOR_TAIL_MARGIN=2 one tailmargin2
one default
} 2>&1 | tee gpurun_out/r5e_ab.txt
bash tools/gpu_timeline.sh; cp gpurun_out/timeline.txt gpurun_out/r5e_step_timeline.txt; cat gpurun_out/timeline.txt
