#!/bin/bash
# round 4, first look at the block solver: small parity variants, then the bench pile against the persistent kernel
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "other_contact_solvers or block_solver" 2>&1 | tail -25 | tee gpurun_out/r4a_pytest.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest 2>&1 | tail -3 | tee gpurun_out/r4a_bench_blocks.log
MI_SOLVER=persist timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest 2>&1 | tail -3 | tee gpurun_out/r4a_bench_persist.log
