"""What the exact seam costs (one GPU): a pile of 2 x 131 072 boxes cut into two x-slabs.
  plain            the single world, default schedule (persistent solver)
  tiling           the single world told the tiling (seam colours first: more colours, same solver)
  jacobi           2 virtual ranks, block-Jacobi seam: both ranks stepped one after the other, one exchange per step
  exact            2 virtual ranks, exact seam: side by side (threads), every sweep one launch + hand-over through the caller's transport (host copies)
  exact_no_peers   ONE rank owning everything in exact mode: the per-sweep launches alone (no messages)
All times are wall clock per step for the WHOLE job on this one GPU (both ranks share it)."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes, sharding

NX = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = scenes.obb_pile(NX, 16, 128)
s = sc.settings()
desc = sharding.tile_grid(sc, 2, 1, 2.5)
SETTLE, TIMED = 240, 40
out = {"bodies": sc.num_bodies}

def timed(step, n=TIMED):
    t0 = time.perf_counter()
    for _ in range(n): step()
    return (time.perf_counter() - t0) / n * 1e3

w = sc.populate(mi.create_world(0))
for _ in range(SETTLE): w.step_fixed(s, sc.dt, 1)
w.counts(); out["plain"] = {"ms_per_step": round(timed(lambda: (w.step_fixed(s, sc.dt, 1), w.counts())), 4), "counts": w.counts()}
w.set_stage_timing(1); w.step_fixed(s, sc.dt, 1); out["plain"]["stage_ms"] = {k: round(v, 4) for k, v in w.stage_times().items()}; w.close()
print("plain", json.dumps(out["plain"]), flush=True)

w = sc.populate(mi.create_world(0)); w.set_seam_tiling(desc)
for _ in range(SETTLE): w.step_fixed(s, sc.dt, 1)
w.counts(); out["tiling"] = {"ms_per_step": round(timed(lambda: (w.step_fixed(s, sc.dt, 1), w.counts())), 4), "counts": w.counts(), "seam": w.seam_stats()}
w.set_stage_timing(1); w.step_fixed(s, sc.dt, 1); out["tiling"]["stage_ms"] = {k: round(v, 4) for k, v in w.stage_times().items()}; w.close()
print("tiling", json.dumps(out["tiling"]), flush=True)

ranks = [sharding.ShardedWorld(sc.populate(mi.create_world(0)), desc, r, "local") for r in range(2)]
for _ in range(SETTLE): sharding.step_local(ranks, s, sc.dt)
out["jacobi"] = {"ms_per_step": round(timed(lambda: sharding.step_local(ranks, s, sc.dt)), 4), "shard": [r.world.shard_counts() for r in ranks],
                 "message_bytes": ranks[0].world.shard_message_bytes()}
print("jacobi", json.dumps(out["jacobi"]), flush=True)
for _ in range(20): sharding.step_local_exact(ranks, s, sc.dt)
out["exact"] = {"ms_per_step": round(timed(lambda: sharding.step_local_exact(ranks, s, sc.dt), 20), 4), "seam": [r.world.seam_stats() for r in ranks],
                "sweep_message_bytes": ranks[0].world.shard_sweep_message_bytes(), "sweeps_per_step": s.num_rigid_solver_iterations,
                "counts": ranks[0].world.counts()}
ranks[0].world.set_stage_timing(1); sharding.step_local_exact(ranks, s, sc.dt)
out["exact"]["rank0_stage_ms_incl_waits"] = {k: round(v, 4) for k, v in ranks[0].world.stage_times().items()}
print("exact", json.dumps(out["exact"]), flush=True)
for r in ranks: r.world.close()

one = sharding.ShardedWorld(sc.populate(mi.create_world(0)), sharding.tile_grid(sc, 1, 1, 2.5), 0, "local")
for _ in range(SETTLE): sharding.step_local([one], s, sc.dt)
out["jacobi_no_peers"] = {"ms_per_step": round(timed(lambda: sharding.step_local([one], s, sc.dt)), 4)}
one.world.shard_set_exact_seam(True, None)
for _ in range(10): sharding.step_local([one], s, sc.dt)
out["exact_no_peers"] = {"ms_per_step": round(timed(lambda: sharding.step_local([one], s, sc.dt)), 4)}
one.world.set_stage_timing(1); sharding.step_local([one], s, sc.dt); out["exact_no_peers"]["stage_ms"] = {k: round(v, 4) for k, v in one.world.stage_times().items()}
print("no peers", json.dumps(out["jacobi_no_peers"]), json.dumps(out["exact_no_peers"]), flush=True)
json.dump(out, open("gpurun_out/exact_seam_cost.json", "w"), indent=1)
