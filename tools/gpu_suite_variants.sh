#!/bin/bash
# the GPU suite under switches that change how steps are enqueued: step graphs forced; sharding / soak tests with block skipping off; resident rows off
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
MI_GRAPH=force timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | tail -6
MI_SHARD_BLOCK_SKIP=0 timeout 900 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_soak.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | tail -4
MI_PERSIST_RESIDENT=0 timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_reference_direct.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | tail -4
