"""Soak of the exact seam on one GPU: a falling, spreading pile in 3 x-slabs for several hundred steps — bodies migrate, manifolds change class, colours are
re-used — exact-seam virtual ranks against the single world told the tiling, compared every step (states of all bodies, bit for bit)."""
import sys, json, time
import numpy as np
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes, sharding, capi

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
out = {}
for name, make, R, margin in (("pile 24x6x12, 3 slabs", lambda: scenes.obb_pile(24, 6, 12, spacing=1.0), 3, 2.5), ("mixed 16x6x16, 2 slabs", lambda: scenes.mixed_stack(16, 6, 16), 2, 2.5),
                              ("ragdolls 6x4, 2 slabs", lambda: scenes.ragdolls(6, 4), 2, 3.5)):
    sc = make()
    desc = sharding.tile_grid(sc, R, 1, margin)
    single = sc.populate(mi.create_world(0)); single.set_seam_tiling(desc)
    ranks = [sharding.ShardedWorld(sc.populate(mi.create_world(0)), desc, r, "local") for r in range(R)]
    s = sc.settings()
    ents = np.flatnonzero(sc.entities["kind"] != capi.ENTITY_STATIC).astype(np.uint32)
    bad = None; migrated = 0; first_owner = None; seam_max = 0; t0 = time.perf_counter()
    for i in range(steps):
        single.step_fixed(s, sc.dt, 1)
        sharding.step_local_exact(ranks, s, sc.dt)
        if sharding.gather_owned(ranks, len(ents)).tobytes() != single.get_body_states(ents).tobytes(): bad = i; break
        seam_max = max(seam_max, single.seam_stats()["seam_manifolds"])
        if i % 20 == 0:
            owner = np.zeros(len(sc.entities), np.int32)
            for r in ranks: owner[r.world.shard_owned_entities()] = r.rank
            if first_owner is None: first_owner = owner
            migrated = max(migrated, int((owner != first_owner).sum()))
    out[name] = {"steps": steps, "first_mismatch_at_step": bad, "bodies": int(len(ents)), "max_seam_manifolds": seam_max, "bodies_that_changed_owner": migrated,
                 "violations": [r.world.seam_stats()["violations"] for r in ranks] + [single.seam_stats()["violations"]], "single_step_modes": single.step_mode_stats(),
                 "seconds": round(time.perf_counter() - t0, 1)}
    print(name, json.dumps(out[name]), flush=True)
    for r in ranks: r.world.close()
    single.close()
json.dump(out, open("gpurun_out/exact_seam_soak.json", "w"), indent=1)
