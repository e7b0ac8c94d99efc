"""Development: what the neighbour exchange costs a rank per step on the library's transport, looped back on one GPU (mi_debug_shard_attach_loopback): rank 0 of a
two-tile scene of 2 x 262 144 boxes (its one neighbour is itself: real records, the real ncclSend / ncclRecv group, unpack, axis).  Prints the step's wall clock with the
exchange, the exchange's own device time (events around it), and the same world's step with the exchange left out (caller's transport, nothing exported).
python tools/gpu_exchange_loopback.py"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes, sharding

sc = scenes.obb_pile(256, 16, 128)
desc = sharding._desc_for(sharding.tile_grid(sc, 2, 1, 2.5), 0)
out = {}
for mode in ("loopback", "no exchange"):
    w = sc.populate(mi.create_world(0)); w.shard_enable(desc)
    if mode == "loopback":
        w.shard_attach_loopback()
    s = sc.settings()
    for _ in range(240): w.step_fixed(s, sc.dt, 1)
    w.shard_exchange_stats(reset=True)
    w.counts(); t0 = time.perf_counter()
    for _ in range(60): w.step_fixed(s, sc.dt, 1)
    w.counts(); dt_ = (time.perf_counter() - t0) / 60
    ex = w.shard_exchange_stats()
    out[mode] = {"ms_per_step": round(dt_ * 1e3, 4), "exchanges": ex["exchanges"], "exchange_device_ms": round(ex["device_ms_sum"] / max(1, ex["exchanges"]), 4),
                 "records_per_exchange": [round(v / max(1, ex["exchanges"]), 1) for v in ex["records_sum"]][:2], "owned": ex["owned_bodies"], "ghosts": ex["ghost_bodies"], "counts": w.counts()}
    print(mode, json.dumps(out[mode]), flush=True)
    w.close()
a, b = out["loopback"], out["no exchange"]
out["exposed_ms"] = round(a["ms_per_step"] - b["ms_per_step"], 4)
out["exposed_fraction_of_exchange_device_time"] = round((a["ms_per_step"] - b["ms_per_step"]) / max(1e-9, a["exchange_device_ms"]), 3)
print(json.dumps({k: out[k] for k in ("exposed_ms", "exposed_fraction_of_exchange_device_time")}))
json.dump(out, open("gpurun_out/exchange_loopback.json", "w"), indent=1)
