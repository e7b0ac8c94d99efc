"""Development: what a rank's step costs as the REPLICATED scene grows (weak scaling without the transport): one GPU plays the middle
tile of 1, 2, 4, 8 x-slabs of a pile 128*R x 16 x 128; the tile itself always holds ~262 144 bodies.  Prints per-stage times."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes, sharding

out = {}
for R in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    sc = scenes.obb_pile(128 * R, 16, 128)
    w = sc.populate(mi.create_world(0))
    if R > 1:
        desc = sharding.tile_grid(sc, R, 1, 2.5)
        sw = sharding.ShardedWorld(w, desc, int(os.environ.get("WEAK_RANK") or R // 2), "local")   # (WEAK_RANK: another tile than the middle one, e.g. 0: the tile at the start of every array)
    s = sc.settings()
    for _ in range(240): w.step_fixed(s, sc.dt, 1)
    w.set_stage_timing(1)
    t = []
    for _ in range(5):
        w.step_fixed(s, sc.dt, 1); t.append(w.stage_times())
    w.set_stage_timing(0)
    for _ in range(5): w.step_fixed(s, sc.dt, 1)
    w.counts(); t0 = time.perf_counter()
    for _ in range(60): w.step_fixed(s, sc.dt, 1)
    w.counts(); dt_ = (time.perf_counter() - t0) / 60
    st = {k: round(float(np.median([x[k] for x in t])), 4) for k in t[0]}
    out[R] = {"bodies_total": sc.num_bodies, "ms_per_step": round(dt_ * 1e3, 4), "counts": w.counts(), "shard": w.shard_counts() if R > 1 else None, "stage_ms": st}
    print(R, json.dumps(out[R]), flush=True)
    w.close()
json.dump(out, open("gpurun_out/exp_weak.json", "w"), indent=1)
