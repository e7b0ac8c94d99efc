#!/bin/bash
# round 3, call A: GPU test suite on the new axis statistic / shard checkpoint / tests, driver-flag bench, self-launched 2-rank functional run, timeline
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r3a_pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest 2>gpurun_out/r3a_bench.err | tail -1 > gpurun_out/r3a_bench_driver_flags.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r3a_bench_driver_flags.json"))
    print("bench", round(d["value"], 1), "contacts", d["config"]["contacts"], "frac", round(d["roofline"]["frac"], 3), "stage", {k: round(v, 3) for k, v in d["stage_ms"].items()}, "timed_ms_total", d.get("timed_ms_total"))
except Exception as e: print("bench FAILED", e)
PY
# the N > 1 path as the driver invokes it, on this ONE-GPU box: a functional check (two processes on one device, caller's transport over gloo)
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --grid 32 16 32 --settle 60 --scaling strong 2>gpurun_out/r3a_two.err | tail -1 > gpurun_out/r3a_two_ranks_one_gpu.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r3a_two_ranks_one_gpu.json"))
    print("two ranks (one GPU, functional):", round(d["value"], 1), d["n_gpus"], [ (r["rank"], r["owned_bodies"], r["ghost_bodies"], round(r["exchange"]["device_ms_per_exchange"], 3), r["exchange"]["records_per_exchange"]) for r in d["per_rank"]])
except Exception as e: print("two ranks FAILED", e); print(open("gpurun_out/r3a_two.err").read()[-1500:])
PY
bash tools/gpu_timeline.sh 2>&1 | tail -2
cp gpurun_out/timeline.txt gpurun_out/r3a_timeline.txt
