"""N > 1 path on CPU: world_size-2 gloo, the sharding / ghost-exchange logic running over the CPU oracle
(tests only; the product passes the HIP library and NCCL=RCCL)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
TILE = (6, 4, 6)
STEPS = 90


def _worker(rank, world_size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, str(ROOT))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    import oracle
    from d3d12renderer_amd.distributed import ShardedWorld
    sw = ShardedWorld(lambda: oracle.create_world(oracle.ORDER_CANONICAL), rank, world_size, dist, tile=TILE, iterations=20, ghost_cols=2)
    s = sw.settings()
    ghost_ok = True
    for i in range(STEPS):
        sw.step(s, sw.dt)
        if i % 30 == 0:
            # ghosts must equal the owners' states right after the exchange: gather every rank's full state table
            mine = sw.owned_states()
            gathered = [torch.zeros(mine.shape, dtype=torch.float32) for _ in range(world_size)]
            dist.all_gather(gathered, torch.from_numpy(mine))
            per_col = TILE[1] * TILE[2]
            if rank > 0:
                got = sw.world.get_body_states(sw.info["ghost_left"])
                ghost_ok &= np.array_equal(got, gathered[rank - 1].numpy()[-2 * per_col:])
            if rank < world_size - 1:
                got = sw.world.get_body_states(sw.info["ghost_right"])
                ghost_ok &= np.array_equal(got, gathered[rank + 1].numpy()[:2 * per_col])
    totals = sw.total_counts()
    np.savez(Path(out_dir) / f"rank{rank}.npz", states=sw.owned_states(), ghost_ok=ghost_ok,
             totals=np.asarray([totals["num_contacts"], totals["num_rigid_bodies"]]), local_contacts=sw.world.counts()["num_contacts"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_pile_matches_single_world(tmp_path, oracle_mod):
    from d3d12renderer_amd import scenes
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    assert bool(r0["ghost_ok"]) and bool(r1["ghost_ok"])
    assert np.array_equal(r0["totals"], r1["totals"])          # all-reduced counts agree on both ranks
    # the same global pen in ONE world (2 tiles wide, no sharding)
    nx, ny, nz = TILE
    sc, info = scenes.obb_pile_tile(0, 1, 2 * nx, ny, nz, ghost_cols=0, solver_iterations=20)
    w = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    w.step_fixed(sc.settings(), sc.dt, STEPS)
    ref = w.get_body_states(np.arange(info["owned"], dtype=np.uint32))
    got = np.concatenate([r0["states"], r1["states"]])
    assert got.shape == ref.shape
    assert np.isfinite(got).all()
    # initial layout identical by construction => same boxes in the same order; the seam is solved Jacobi-style,
    # so trajectories agree to a tolerance, not bit-for-bit
    err = np.abs(got[:, :3] - ref[:, :3]).max(axis=1)
    assert np.median(err) < 2e-2
    assert np.quantile(err, 0.95) < 0.25
    ref_contacts = w.counts()["num_contacts"]
    owned_total = 2 * nx * ny * nz
    assert int(r0["totals"][1]) == owned_total + 2 * 2 * ny * nz + 0   # owned + ghost copies (2 columns each side of the seam)
    # contacts: each rank also counts its ghost-side contacts, so the sum exceeds the single-world count slightly
    assert 0.9 * ref_contacts < int(r0["totals"][0]) < 1.5 * ref_contacts


def test_tile_scene_is_consistent_with_global_scene():
    from d3d12renderer_amd import scenes
    nx, ny, nz = 4, 3, 5
    glob, ginfo = scenes.obb_pile_tile(0, 1, 3 * nx, ny, nz, ghost_cols=0)
    per_col = ny * nz
    for tile in range(3):
        sc, info = scenes.obb_pile_tile(tile, 3, nx, ny, nz, ghost_cols=1)
        own = sc.entities[: info["owned"]]
        assert np.array_equal(own, glob.entities[tile * nx * per_col: (tile + 1) * nx * per_col])
        assert np.array_equal(sc.colliders[: info["owned"]], glob.colliders[tile * nx * per_col: (tile + 1) * nx * per_col])
        if tile > 0:
            assert np.array_equal(sc.entities[info["ghost_left"]], glob.entities[(tile * nx - 1) * per_col: tile * nx * per_col])
            assert len(info["send_left"]) == per_col
        if tile < 2:
            assert np.array_equal(sc.entities[info["ghost_right"]], glob.entities[(tile + 1) * nx * per_col: ((tile + 1) * nx + 1) * per_col])
            assert len(info["send_right"]) == per_col
