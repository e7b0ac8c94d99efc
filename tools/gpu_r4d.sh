#!/bin/bash
# round 4: where the block solver's time goes — an extra, knocked-out launch of the solver per step (the step's results stay right)
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
rm -f gpurun_out/r4d_knockouts.log
for d in 0 1 2 3 4 5 7 8 16 23 31; do
  MI_BLOCK_DBG=$d timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-at-rest 2>&1 >/dev/null | grep "knock-outs" | tail -2 | tee -a gpurun_out/r4d_knockouts.log
done
