#!/bin/bash
# scratch: run selected GPU tests
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
timeout 600 python -m pytest tests/test_scene_formats.py -x -q -m gpu 2>&1 | tail -40 | tee gpurun_out/two.log
