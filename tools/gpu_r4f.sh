#!/bin/bash
# round 4: knock-out launches of the block solver at the settled bench state only (after step 250), with per-phase wall-clock stamps
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
MI_BLOCK_DBG_AFTER=250 MI_BLOCK_DBG=${DBGLIST:-0,15,1,8} timeout 400 python bench.py --steps ${DBGSTEPS:-90} --warmup 5 --no-cpu-baseline --no-at-rest 2> gpurun_out/r4f.err > gpurun_out/r4f.out
grep "knock-outs\|phases" gpurun_out/r4f.err | tee gpurun_out/r4f_knockouts.log
