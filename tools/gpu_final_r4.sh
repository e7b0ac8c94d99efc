#!/bin/bash
# round 4, evidence run at HEAD: full GPU suite; the direct-against-the-reference harness with its printed deviations; rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE
# passes (tools/gpu_pmc.sh) and the default bench (with at_rest + cpu_baseline); the driver's flags; bench --pmc; step timelines; the block solver's line;
# the host-boundary rates (pose rows / per-array path); the other configs; the learning DLL's throughput
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/fin_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/fin_pytest.log
grep -E "passed|failed|rc=" gpurun_out/fin_pytest.log | tail -3
timeout 600 python -m pytest tests/test_gpu_reference_direct.py -q -m gpu -s 2>&1 | grep -E "teacher-forced|replay|passed|failed" > gpurun_out/fin_teacher_forced.txt; tail -2 gpurun_out/fin_teacher_forced.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/gpu_pmc.sh 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/fin_bench_driver_flags.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest --pmc 2>gpurun_out/fin_pmc_err.log | tail -1 > gpurun_out/fin_bench_pmc.json
MI_SOLVER=blocks timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest 2>/dev/null | tail -1 > gpurun_out/fin_bench_blocks.json
python - <<'PY'
import json
for f in ("bench_default", "fin_bench_driver_flags", "fin_bench_pmc", "fin_bench_blocks"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "contacts", d["config"]["contacts"], "frac", round(d["roofline"]["frac"], 3), "of achievable", round(d["roofline"]["frac_of_achievable"], 3),
              "launch us", round(d["roofline"]["avg_launch_us"], 1), "kind", d.get("solver_kind"), "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_source"),
              "step frac", round(d["roofline"]["whole_step"]["frac"], 3), "at_rest", d.get("at_rest", {}).get("value"), "reruns", d["step_modes_timed"]["synchronous_reruns"])
        if "cpu_baseline" in d: print("  cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "scalar_1core", "avx2_1core")})
    except Exception as e: print(f, "FAILED", e)
PY
bash tools/gpu_timeline.sh 2>&1 | tail -1; cp gpurun_out/timeline.txt gpurun_out/fin_step_timeline.txt
TL_EXTRA="--settle 1500" bash tools/gpu_timeline.sh 2>&1 | tail -1; cp gpurun_out/timeline.txt gpurun_out/fin_step_timeline_at_rest.txt
timeout 400 python tools/gpu_pcie_rate.py > gpurun_out/fin_host_boundary_rate.json 2> gpurun_out/fin_host_boundary_rate.err; cut -c1-600 gpurun_out/fin_host_boundary_rate.json
MI_POSE_STREAM=0 timeout 400 python tools/gpu_pcie_rate.py > gpurun_out/fin_host_boundary_rate_per_array_path.json 2>> gpurun_out/fin_host_boundary_rate.err
bash tools/gpu_cfgs.sh 2>&1 | tail -9 | cut -c1-200
cp gpurun_out/cfgs.json gpurun_out/fin_other_configs.json
bash tools/gpu_learning.sh 2>&1 | tail -5 | cut -c1-200; cp gpurun_out/learning.json gpurun_out/fin_learning.json
bash tools/gpu_two.sh 2>&1 | tail -6 | cut -c1-300
timeout 600 python tools/exp_weak.py 1 8 > gpurun_out/fin_weak.log 2>&1; tail -2 gpurun_out/fin_weak.log | cut -c1-200
