#!/bin/bash
# terrain: the counting pass stashes its contacts, the WRITE pass copies them — parity (terrain tests), timing
ulimit -c 0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sharding.py -q -m gpu -x -k "terrain or heightmap or zoo or edge" > gpurun_out/r3ab_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3ab_pytest.log
tail -3 gpurun_out/r3ab_pytest.log
CFGS=terrain bash tools/gpu_cfgs.sh 2>&1 | grep terrain | cut -c1-700
MI_HM_STASH=0 CFGS=terrain bash tools/gpu_cfgs.sh 2>&1 | grep terrain | cut -c1-700
