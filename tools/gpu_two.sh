#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
run() { timeout 120 python bench.py --steps 30 --warmup 250 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['stage_ms']['solve'],4), d['config']['contacts'])
except Exception as e: print('$1 failed', e)"; }
MI_ASYNC=0 MI_GVEL_ALLOC=plain MI_PHYSICS_LIB=d3d12renderer_amd/libmi_b.so run nowait_plainmem_plainops_persist
MI_ASYNC=0 MI_PHYSICS_LIB=d3d12renderer_amd/libmi_b.so run nowait_ucmem_plainops_persist
MI_ASYNC=0 MI_GVEL_ALLOC=plain MI_PHYSICS_LIB=d3d12renderer_amd/libmi_a.so run nowait_plainmem_sc1_persist
