#!/bin/bash
# dev helper: the driver-flag bench with every build_exp/libmi_physics_*.so (and the in-tree library before, between and after): steps/s, solver launch, stage times
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest"
one() { timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), 'steps/s solver', round(d['roofline']['avg_launch_us'],1), {k[:5]:round(v,3) for k,v in d['stage_ms'].items()})"; }
unset MI_PHYSICS_LIB; one in-tree
for lib in ${VARIANTS:-$(ls build_exp/libmi_physics_*.so)}; do
  export MI_PHYSICS_LIB=$PWD/$lib; one $(basename $lib .so | sed 's/libmi_physics_//')
done
unset MI_PHYSICS_LIB; one in-tree
