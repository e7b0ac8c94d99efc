#!/bin/bash
# round 4: block solver at the bench size — bench (blocks / persist), kernel timeline of one settled step, small parity variants
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
MI_BLOCK_DEBUG=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest > gpurun_out/r4c_bench_blocks.log 2> gpurun_out/r4c_bench_blocks.err; echo "rc=$?" >> gpurun_out/r4c_bench_blocks.err
tail -1 gpurun_out/r4c_bench_blocks.log | cut -c1-600; tail -3 gpurun_out/r4c_bench_blocks.err | cut -c1-500
TL_EXTRA="" bash tools/gpu_timeline.sh; cp gpurun_out/timeline.txt gpurun_out/r4c_timeline_blocks.txt; cat gpurun_out/r4c_timeline_blocks.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "other_contact_solvers or block_solver or bench_size_solvers" > gpurun_out/r4c_pytest.log 2>&1
tail -8 gpurun_out/r4c_pytest.log
