#!/bin/bash
# dev helper (round 2): GPU tests + default bench + the driver's flags + rocprofv3 kernel stats, one gpurun call
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
timeout 600 python bench.py 2>gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default.json; tail -c 600 gpurun_out/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest 2>&1 | tail -1 > gpurun_out/bench_driver_flags.json
python - <<'PY'
import json
for f in ("bench_default", "bench_driver_flags"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "value", round(d["value"], 1), "contacts", d["config"]["contacts"], "frac", round(d["roofline"]["frac"], 3), "step frac", round(d["roofline"]["whole_step"]["frac"], 3),
              "at_rest", d.get("at_rest", {}).get("value"), "stage", {k: round(v, 3) for k, v in d["stage_ms"].items()})
        if "cpu_baseline" in d: print(" cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "scalar_1core", "avx2_1core")})
    except Exception as e: print(f, "FAILED", e)
PY
RAW=/tmp/prof_raw; rm -rf $RAW; mkdir -p $RAW
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o r02 -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-at-rest > gpurun_out/stats_bench.log 2>&1
python tools/summarize_prof.py $RAW gpurun_out/prof_summary 2>&1 | tail -5
