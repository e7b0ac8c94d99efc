#!/bin/bash
# round 3, call U: full GPU suite on HEAD (after the exact seam)
ulimit -c 0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r3u_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3u_pytest.log
tail -12 gpurun_out/r3u_pytest.log
