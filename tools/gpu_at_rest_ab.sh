export TMPDIR=/tmp; ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "other_contact_solvers or bench_size" 2>&1 | tail -2
for V in 0 1; do
MI_PERSIST_RESIDENT=$V timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/rest_$V.json
python - <<PY
import json
d = json.load(open("gpurun_out/rest_$V.json")); a = d["at_rest"]
print("resident=$V bench", round(d["value"], 1), "solver", round(d["roofline"]["avg_launch_us"], 1), "| at rest", round(a["value"], 1), "ms", round(a["ms_per_step"], 4), "solver us", round(a["solver_avg_launch_us"], 1), "contacts", a["contacts"])
PY
done
