#!/bin/bash
# same-box A/B of development builds (tools/build_variant.py NAME ...) against the in-tree library, at the driver's flags: bash tools/gpu_ab_variants.sh NAME [NAME ...]
# (REPS=n: alternations, default 2).  Prints steps/s, the solver launch, and the stage times of the 3-step sample.
export TMPDIR=/tmp
one() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('$1'.ljust(10), round(d['value'],1), 'solver', round(d['roofline']['avg_launch_us'],1), ' '.join(f'{k[:6]} {v*1e3:.1f}' for k,v in s.items() if k not in ('total','solve')))"; }
for r in $(seq ${REPS:-2}); do
  one base
  for v in "$@"; do MI_PHYSICS_LIB=build_exp/libmi_physics_$v.so one $v; done
done
