#!/bin/bash
ulimit -c 0
mkdir -p gpurun_out
run() { timeout 120 python bench.py --steps 30 --warmup 250 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['stage_ms']['solve'],4))
except Exception as e: print('$1 failed', e)"; }
MI_SOLVER=persist MI_PERSIST_WAVES=512 run p512
MI_SOLVER=persist MI_PERSIST_WAVES=768 run p768
MI_SOLVER=persist MI_PERSIST_WAVES=896 run p896
MI_SOLVER=persist MI_PERSIST_WAVES=1024 run p1024
MI_FLOW_LDS=70000 run flow_lds70k
