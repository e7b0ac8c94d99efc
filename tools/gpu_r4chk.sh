#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "trajectory_and_contacts" > gpurun_out/r4chk_pytest.log 2>&1; tail -12 gpurun_out/r4chk_pytest.log | cut -c1-300
