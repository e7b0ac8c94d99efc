#!/bin/bash
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
export TMPDIR=/tmp
ulimit -c 0
cat > /tmp/pile.py <<'PY'
import sys, time, os, hashlib
sys.path.insert(0, os.getcwd())
import torch; torch.cuda.init()
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
nx, ny, nz, steps = map(int, sys.argv[1:5])
sc = scenes.obb_pile(nx, ny, nz); w = sc.populate(mi.create_world(0)); s = sc.settings()
t0 = time.time()
for i in range(steps): w.step_fixed(s, sc.dt, 1)
p, q = w.physics_transforms()
print(os.environ.get("TAG"), "ms/step", round((time.time() - t0) / steps * 1e3, 3), "kind", w.solver_kind(), "modes", w.step_mode_stats(), "contacts", w.counts()["num_contacts"], "sha", hashlib.sha1(p.tobytes() + q.tobytes()).hexdigest()[:12], flush=True)
PY
TAG=blocks-64k timeout 100 python /tmp/pile.py 64 16 64 260 2>&1 | tail -1 | tee gpurun_out/r4w_pile.log
TAG=persist-64k MI_SOLVER=persist timeout 100 python /tmp/pile.py 64 16 64 260 2>&1 | tail -1 | tee -a gpurun_out/r4w_pile.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "other_contact_solvers or block_solver or bench_size_solvers" > gpurun_out/r4w_pytest.log 2>&1; tail -6 gpurun_out/r4w_pytest.log
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest > gpurun_out/r4w_bench.log 2> gpurun_out/r4w_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4w_bench.log").read().strip().split("\n")[-1])
print("bench", round(d["value"], 1), "steps/s", "solver us", round(d["roofline"]["avg_launch_us"], 1), "kind", d["solver_kind"], {k: round(v * 1e3, 1) for k, v in d["stage_ms"].items()}, d["step_modes_timed"]["synchronous_reruns"])
PY
