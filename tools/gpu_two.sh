#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_capi_symbols.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/two.log
