#!/bin/bash
# dev helper: the solver-variant parity tests on the GPU box
ulimit -c 0
mkdir -p gpurun_out
cd oracle && make >/dev/null 2>&1; cd ..
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "other_contact_solvers" 2>&1 | grep -E "^E|passed|failed" | head -12
