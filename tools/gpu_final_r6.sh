#!/bin/bash
# round 6, evidence run at HEAD: full GPU suite; the direct-against-the-reference harness with its printed deviations and the dataflow replay; smoke; rocprofv3 kernel stats +
# FETCH_SIZE / WRITE_SIZE passes and the default bench (with at_rest + cpu_baseline); the driver's flags twice; bench --pmc; resident rows off on the same box; step timelines;
# the persistent solver's visit stamps and the knock-out harness; the host-boundary rates; the other configs; the learning DLL's throughput; two ranks on one GPU; what a
# rank pays for the replicated scene; the exchange looped back; a soak with step graphs forced
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/fin_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/fin_pytest.log
grep -E "passed|failed|rc=" gpurun_out/fin_pytest.log | tail -3
timeout 600 python -m pytest tests/test_gpu_reference_direct.py -q -m gpu -s 2>&1 | grep -E "teacher-forced|dataflow replay|replay|passed|failed" > gpurun_out/fin_teacher_forced.txt; tail -2 gpurun_out/fin_teacher_forced.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "suite + smoke at $(( $(date +%s) - T0 )) s"
bash tools/gpu_pmc.sh 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/fin_bench_driver_flags.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest --pmc 2>gpurun_out/fin_pmc_err.log | tail -1 > gpurun_out/fin_bench_pmc.json
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest"
MI_PERSIST_RESIDENT=0 timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/fin_bench_no_resident_rows.json
timeout 300 $B 2>/dev/null | tail -1 > gpurun_out/fin_bench_driver_flags_again.json
python - <<'PY'
import json
for f in ("bench_default", "fin_bench_driver_flags", "fin_bench_pmc", "fin_bench_no_resident_rows", "fin_bench_driver_flags_again"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "contacts", d["config"]["contacts"], "frac", round(d["roofline"]["frac"], 3), "bound", d["roofline"]["bound"],
              "launch us", round(d["roofline"]["avg_launch_us"], 1), "hop us", round(d["roofline"]["chain"]["us_per_hop"], 3), "kind", d.get("solver_kind"), "traffic", d["roofline"]["traffic"], d["roofline"].get("traffic_source"),
              "traffic frac", d["roofline"]["traffic_frac"], "step frac", round(d["roofline"]["whole_step"]["frac"], 3), "at_rest", d.get("at_rest", {}).get("value"), "reruns", d["step_modes_timed"]["synchronous_reruns"],
              "l2", (d.get("with_whole_step_events") or {}).get("value"))
        if "cpu_baseline" in d: print("  cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "scalar_1core", "avx2_1core")})
    except Exception as e: print(f, "FAILED", e)
PY
echo "benches at $(( $(date +%s) - T0 )) s"
bash tools/gpu_timeline.sh 2>&1 | tail -1; cp gpurun_out/timeline.txt gpurun_out/fin_step_timeline.txt
TL_EXTRA="--settle 1500" bash tools/gpu_timeline.sh 2>&1 | tail -1; cp gpurun_out/timeline.txt gpurun_out/fin_step_timeline_at_rest.txt
bash tools/gpu_timeline2.sh > gpurun_out/fin_solver_visit_stamps.txt 2>&1; tail -3 gpurun_out/fin_solver_visit_stamps.txt
KO_SETTLES="245" bash tools/gpu_knockout.sh > gpurun_out/fin_knockout.txt 2>&1; grep -c knockout gpurun_out/fin_knockout.txt
timeout 400 python tools/gpu_pcie_rate.py > gpurun_out/fin_host_boundary_rate.json 2> gpurun_out/fin_host_boundary_rate.err; cut -c1-600 gpurun_out/fin_host_boundary_rate.json
bash tools/gpu_cfgs.sh 2>&1 | tail -9 | cut -c1-200
cp gpurun_out/cfgs.json gpurun_out/fin_other_configs.json
bash tools/gpu_learning.sh 2>&1 | tail -5 | cut -c1-200; cp gpurun_out/learning.json gpurun_out/fin_learning.json
bash tools/gpu_two.sh 2>&1 | tail -6 | cut -c1-300
timeout 600 python tools/exp_weak.py 1 8 > gpurun_out/fin_weak.log 2>&1; tail -2 gpurun_out/fin_weak.log | cut -c1-200
timeout 600 python tools/gpu_exchange_loopback.py > gpurun_out/fin_exchange_loopback.log 2>&1; tail -1 gpurun_out/fin_exchange_loopback.log
timeout 300 python tools/gpu_sync_vs_spec.py > gpurun_out/fin_sync_vs_spec.txt 2>&1; tail -1 gpurun_out/fin_sync_vs_spec.txt
MI_GRAPH=force timeout 500 python tools/gpu_soak.py 2>/dev/null | tail -1 > gpurun_out/fin_soak.json; cut -c1-400 gpurun_out/fin_soak.json
echo "all done at $(( $(date +%s) - T0 )) s"
