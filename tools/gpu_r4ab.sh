#!/bin/bash
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "pose or accumulator" > gpurun_out/r4ab_pytest.log 2>&1; tail -8 gpurun_out/r4ab_pytest.log
timeout 300 python tools/gpu_pcie_rate.py > gpurun_out/r4ab_rate.json 2> gpurun_out/r4ab_rate.err; cat gpurun_out/r4ab_rate.json; tail -3 gpurun_out/r4ab_rate.err
MI_POSE_STREAM=0 timeout 300 python tools/gpu_pcie_rate.py > gpurun_out/r4ab_rate_off.json 2> gpurun_out/r4ab_rate_off.err; cat gpurun_out/r4ab_rate_off.json; tail -3 gpurun_out/r4ab_rate_off.err
