"""Development: exact seam on one GPU — virtual ranks (threads) vs the single world told the tiling vs the oracle told the tiling."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
import oracle; oracle.build()
from d3d12renderer_amd import scenes, sharding, capi

def run(kind, R, steps):
    sc = scenes.obb_pile(12, 4, 8, spacing=1.0) if kind == "pile" else scenes.ragdolls(4, 3)
    desc = sharding.tile_grid(sc, R, 1, 2.5 if kind == "pile" else 3.5)
    single = sc.populate(mi.create_world(0)); single.set_seam_tiling(desc)
    ora = sc.populate(oracle.create_world(oracle.ORDER_CANONICAL)); ora.set_seam_tiling(desc)
    ranks = [sharding.ShardedWorld(sc.populate(mi.create_world(0)), desc, r, "local") for r in range(R)]
    s = sc.settings()
    ents = np.flatnonzero(sc.entities["kind"] != capi.ENTITY_STATIC).astype(np.uint32)
    for i in range(steps):
        single.step_fixed(s, sc.dt, 1); ora.step_fixed(s, sc.dt, 1)
        sharding.step_local_exact(ranks, s, sc.dt)
        got = sharding.gather_owned(ranks, len(ents)); ref = single.get_body_states(ents); o = ora.get_body_states(ents)
        if ref.tobytes() != o.tobytes():
            print(kind, R, "step", i, "GPU single-with-tiling != oracle", np.abs(ref - o).max(), single.counts(), ora.counts(), single.seam_stats(), ora.seam_stats()); return
        if got.tobytes() != ref.tobytes():
            d = np.abs(got - ref).max(axis=1)
            print(kind, R, "step", i, "ranks != single", (d > 0).sum(), d.max(), [r.world.seam_stats() for r in ranks], single.seam_stats()); return
    print(kind, R, "OK", steps, single.seam_stats(), [r.world.seam_stats() for r in ranks], single.counts()["num_colors"])
run("pile", 2, 40); run("pile", 3, 40); run("ragdolls", 2, 40)
