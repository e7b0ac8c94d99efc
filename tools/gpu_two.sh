#!/bin/bash
# functional check of the N > 1 bench path on ONE GPU (both ranks share it: the caller's transport over torch.distributed; the library's RCCL transport wants a device per rank):
# bench.py launches its ranks itself; block-Jacobi seam, then the exact seam
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --grid 64 8 64 --no-cpu-baseline 2> gpurun_out/two_ranks.err | tail -1 > gpurun_out/two_ranks.json
timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --grid 64 8 64 --no-cpu-baseline --seam exact 2> gpurun_out/two_ranks_exact.err | tail -1 > gpurun_out/two_ranks_exact.json
python - <<'PY'
import json
for f in ("two_ranks", "two_ranks_exact"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, "value", round(d["value"], 1), "n_gpus", d["n_gpus"], "ms", round(d["ms_per_step"], 3), "|", d["config"]["sharding"][:260])
        for r in d.get("per_rank", [])[:2]: print("   rank", {k: r[k] for k in list(r)[:8]})
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/{f}.err").read()[-1500:])
PY
