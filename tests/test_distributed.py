"""Sharded worlds (include/mi_shard.h) without a GPU: the CPU oracle mirrors the product's sharding (oracle/ora_world.cpp, "sharded
world"), so the N > 1 path — ownership by position, ghosts, migration, packing / unpacking of the neighbour messages, the owner
rule — is exercised over gloo with real processes and compared, bit for bit, with the same ranks run one after the other in one
process.  The GPU versions of these tests are in tests/test_gpu_sharding.py.
"""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from d3d12renderer_amd import capi, scenes, sharding

ROOT = Path(__file__).resolve().parent.parent
STEPS = 80


def _scene(kind="pile"):
    if kind == "terrain":
        return scenes.terrain_field(8, 2, 8, with_unsupported=False)
    if kind == "lopsided":
        return scenes.obb_pile(16, 3, 8, spacing=1.0)
    return scenes.ragdolls(4, 3) if kind == "ragdolls" else scenes.obb_pile(12, 4, 8, spacing=1.0)


MARGIN = {"pile": 2.5, "ragdolls": 3.5, "terrain": 2.5, "lopsided": 1.5}      # islands are classified by their root body: the margin has to cover an island's reach
REBALANCE_EVERY = 8


def _grid(sc, kind, num_ranks, tiles_z):
    """The tile grid of a test scene.  "lopsided": the pile under a grid that was laid out for something else — the first column of tiles holds
    nothing, the last one most of the pile — so that the load balance has work to do."""
    desc = sharding.tile_grid(sc, num_ranks, tiles_z, MARGIN[kind])
    if kind == "lopsided":
        desc.origin_x -= desc.tile_size_x
    return desc


def _virtual_ranks(make_world, sc, num_ranks, tiles_z=1, margin=2.5, kind=None):
    desc = _grid(sc, kind, num_ranks, tiles_z) if kind else sharding.tile_grid(sc, num_ranks, tiles_z, margin)
    return [sharding.ShardedWorld(sc.populate(make_world()), desc, r, "local") for r in range(num_ranks)], desc


def _worker(rank, world_size, port, out_dir, tiles_z, kind="pile", transport="dist"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, str(ROOT))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    import oracle
    sc = _scene(kind)
    desc = _grid(sc, kind, world_size, tiles_z)
    sw = sharding.ShardedWorld(sc.populate(oracle.create_world(oracle.ORDER_CANONICAL)), desc, rank, transport, dist)
    assert sw.transport == "dist" and (transport == "dist" or "unavailable" in sw.note)
    s = sc.settings()
    owned_per_step = []
    for i in range(STEPS):
        sw.step(s, sc.dt)
        owned_per_step.append(sw.world.shard_counts()["owned_bodies"])
        if kind == "lopsided" and i % REBALANCE_EVERY == REBALANCE_EVERY - 1:
            sw.rebalance()                                      # histograms all-reduced over gloo, borders moved for the next step
    ents, st = sw.owned_states()
    np.savez(Path(out_dir) / f"rank{rank}.npz", ents=ents, states=st, owned=np.asarray(owned_per_step), counts=np.asarray(list(sw.world.shard_counts().values())),
             borders=np.concatenate(sw.world.shard_get_borders(desc.tiles_x, desc.tiles_z)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world_size,tiles_z,kind,transport", [(2, 1, "pile", "dist"), (4, 2, "pile", "dist"), (2, 1, "ragdolls", "dist"), (2, 1, "pile", "rccl"), (2, 1, "terrain", "dist"),
                                                               (3, 1, "lopsided", "dist")],
                         ids=["2 ranks, x slabs", "4 ranks, 2 x 2 tiles", "2 ranks, ragdolls", "2 ranks, library transport unavailable -> caller's transport", "2 ranks, heightmap terrain",
                              "3 ranks, borders rebalanced every 8 steps"])
def test_processes_over_gloo_equal_virtual_ranks_bit_for_bit(tmp_path, oracle_mod, world_size, tiles_z, kind, transport):
    """R processes exchanging the neighbour messages over gloo == R worlds of one process with the messages copied by hand:
    the transport carries exactly what the library packed, nothing depends on timing or on who runs a tile.  Last case: the ranks
    ask for the library's own RCCL transport, which the CPU oracle does not have — all of them agree to fall back (sharding.py)."""
    port = 29500 + (os.getpid() % 2000) + world_size
    mp.spawn(_worker, args=(world_size, port, str(tmp_path), tiles_z, kind, transport), nprocs=world_size, join=True)
    sc = _scene(kind)
    ranks, desc = _virtual_ranks(lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL), sc, world_size, tiles_z, kind=kind)
    s = sc.settings()
    owned = []
    for i in range(STEPS):
        sharding.step_local(ranks, s, sc.dt)
        owned.append([r.world.shard_counts()["owned_bodies"] for r in ranks])
        if kind == "lopsided" and i % REBALANCE_EVERY == REBALANCE_EVERY - 1:
            sharding.rebalance_local(ranks)
    owned = np.asarray(owned)
    if kind == "lopsided":
        assert owned[0].max() > 0.6 * sc.num_bodies and owned[-1].max() < 0.42 * sc.num_bodies, f"load balance: {owned[0]} -> {owned[-1]}"
    assert (owned.sum(axis=1) == sc.num_bodies).all(), "every body has exactly one owner in every step"
    for r in range(world_size):
        got = np.load(tmp_path / f"rank{r}.npz")
        ents, st = ranks[r].owned_states()
        assert np.array_equal(got["ents"], ents) and got["states"].tobytes() == st.tobytes(), f"rank {r}"
        assert np.array_equal(got["owned"], owned[:, r])
        assert np.array_equal(got["counts"], np.asarray(list(ranks[r].world.shard_counts().values())))
        assert got["borders"].tobytes() == np.concatenate(ranks[r].world.shard_get_borders(desc.tiles_x, desc.tiles_z)).tobytes()


def test_migration_ghosts_and_owner_rule(oracle_mod):
    """A pile that spreads across three x tiles: bodies change owner (migration), every step the owned sets partition the bodies,
    a ghost's state on the neighbour equals its owner's state after the exchange, and the owner rule counts every manifold once
    (owned manifolds summed over ranks == the union of the tiles' manifold sets)."""
    sc = scenes.obb_pile(9, 5, 6, spacing=0.9)
    ranks, desc = _virtual_ranks(lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL), sc, 3, 1, margin=2.0)
    s = sc.settings()
    first_owner = None; changed = 0
    for i in range(120):
        sharding.step_local(ranks, s, sc.dt)
        owner = np.full(len(sc.entities), -1)
        for r in ranks:
            e = r.world.shard_owned_entities()
            assert (owner[e] == -1).all(); owner[e] = r.rank
        assert (owner[: sc.num_bodies] >= 0).all()
        if first_owner is None:
            first_owner = owner.copy()
        changed = max(changed, int((owner != first_owner).sum()))
        if i % 20 == 0:      # ghosts == owners right after the exchange
            truth = sharding.gather_owned(ranks, sc.num_bodies)
            for r in ranks:
                states = r.world.get_body_states(np.arange(sc.num_bodies, dtype=np.uint32))
                cog_x = states[:, 0]
                x0 = desc.origin_x + r.world.L.shard_tile_of_rank(3, 1, r.rank) * desc.tile_size_x
                near = (cog_x > x0 - 0.5 * desc.ghost_margin) & (cog_x < x0 + desc.tile_size_x + 0.5 * desc.ghost_margin)
                assert states[near].tobytes() == truth[near].tobytes(), f"step {i} rank {r.rank}: a body inside the extended tile is stale"
    assert changed > 0, "no body ever changed owner: the scene does not test migration"
    total = sum(r.world.shard_counts()["owned_contacts"] for r in ranks)
    assert total > sc.num_bodies, "a settled pile has more than one contact per body"


def test_sharded_pile_stays_close_to_the_single_world(oracle_mod, record_property):
    """What the block-Jacobi seam costs: the sharded pile versus the unsharded world (same scene, same steps).  Chaotic pile, so
    only statistics are comparable — reported, and bounded loosely: nothing explodes, nothing falls through, the contact count of
    the whole scene agrees within 10 %."""
    sc = _scene()
    ranks, _ = _virtual_ranks(lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL), sc, 2)
    single = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings()
    horizon = {}
    for i in range(160):
        sharding.step_local(ranks, s, sc.dt); single.step_fixed(s, sc.dt, 1)
        if i + 1 in (1, 10, 40, 80, 160):
            sharded = sharding.gather_owned(ranks, sc.num_bodies)
            ref = single.get_body_states(np.arange(sc.num_bodies, dtype=np.uint32))
            err = np.linalg.norm(sharded[:, :3] - ref[:, :3], axis=1)
            horizon[i + 1] = {"max_m": float(err.max()), "median_m": float(np.median(err)), "bodies_differing": int((err > 0).sum())}
    contacts = sum(r.world.shard_counts()["owned_contacts"] for r in ranks)
    out = {"median_position_error_m": float(np.median(err)), "p95_position_error_m": float(np.percentile(err, 95)),
           "contacts_sharded": int(contacts), "contacts_single": int(single.counts()["num_contacts"]), "position_error_after_steps": horizon}
    assert horizon[1]["max_m"] < 5e-3, "one step of block Jacobi against one step of Gauss-Seidel across the seam: millimetres at most"
    record_property("sharded_vs_single", str(out)); print("sharded vs single world:", out)
    assert np.isfinite(sharded).all() and sharded[:, 1].min() > -0.05
    assert abs(contacts - single.counts()["num_contacts"]) <= 0.1 * single.counts()["num_contacts"]
    assert np.median(err) < 0.25


def test_articulated_islands_stay_on_one_rank(oracle_mod):
    """cfg4-style scene in three x tiles: every ragdoll (14 bodies, 13 joints) is owned, ghosted or ignored as ONE — its root body's
    centre decides — so no joint ever spans ranks; ragdolls that tumble across a border migrate whole."""
    sc = scenes.ragdolls(6, 2)
    ranks, _ = _virtual_ranks(lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL), sc, 3, 1, margin=3.5)
    s = sc.settings()
    rng = np.random.default_rng(5)
    for r in ranks:                                             # a sideways shove so that some ragdolls cross a tile border
        r.world.apply_forces(np.arange(0, sc.num_bodies, 14, dtype=np.uint32), np.tile([[9000.0, 0.0, 0.0]], (sc.num_bodies // 14, 1)))
    first = None; moved = 0
    for i in range(150):
        sharding.step_local(ranks, s, sc.dt)
        owner = np.full(sc.num_bodies, -1)
        for r in ranks:
            e = r.world.shard_owned_entities(); e = e[e < sc.num_bodies]
            assert (owner[e] == -1).all(); owner[e] = r.rank
        assert (owner >= 0).all()
        per_doll = owner.reshape(-1, 14)
        assert (per_doll == per_doll[:, :1]).all(), f"step {i}: a ragdoll is split between ranks"
        if first is None:
            first = per_doll[:, 0].copy()
        moved = max(moved, int((per_doll[:, 0] != first).sum()))
    assert moved > 0, "no ragdoll ever changed rank"
    st = sharding.gather_owned(ranks, sc.num_bodies)
    assert np.isfinite(st).all() and st[:, 1].min() > -0.2


def test_sharded_independent_islands_equal_the_single_world(oracle_mod):
    """cfg4-style scene ("independent units"): ragdolls that touch the ground but not each other.  Block Jacobi across the seam then
    has nothing to approximate, and the schedule of an island is graph-local (priorities, colour history and joint order do not look past the
    island): the sharded result IS the single world's, bit for bit, every step — as long as ONE global quantity agrees, the sweep axis.  It is
    chosen from the variance of the colliders a world simulates and orients pairs of EQUAL shape type the way the reference's sweep emits them
    (collision_narrow.cpp:2374): a rank that sees a quarter of a scene may pick another axis and orient a capsule-capsule or gear-tooth pair the
    other way round (same contact, A and B swapped: last-bit differences from there on).  Here the axes agree throughout, which is asserted."""
    sc = scenes.ragdolls(4, 3, spacing=4.0)
    ranks, _ = _virtual_ranks(lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL), sc, 2, 1, margin=3.5)
    single = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    s = sc.settings(); ids = np.arange(sc.num_bodies, dtype=np.uint32)
    for i in range(150):
        sharding.step_local(ranks, s, sc.dt); single.step_fixed(s, sc.dt, 1)
        assert {r.world.counts()["sorting_axis"] for r in ranks} == {single.counts()["sorting_axis"]}, f"step {i}: the precondition of this test"
        assert sharding.gather_owned(ranks, sc.num_bodies).tobytes() == single.get_body_states(ids).tobytes(), f"step {i}"
        assert sum(r.world.shard_counts()["owned_contacts"] for r in ranks) == single.counts()["num_contacts"]
    assert single.counts()["num_contacts"] > 100


def test_shard_api_rejects_what_it_cannot_do(oracle_mod):
    sc = _scene()
    w = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    bad = sharding._desc_for(sharding.tile_grid(sc, 2), 0); bad.num_ranks = 3
    with pytest.raises(capi.PhysicsError):
        w.shard_enable(bad)
    bad = sharding._desc_for(sharding.tile_grid(sc, 2), 0); bad.ghost_margin = 1e9
    with pytest.raises(capi.PhysicsError):
        w.shard_enable(bad)
    assert [w.L.shard_tile_of_rank(2, 2, r) for r in range(4)] == [0, 1, 2, 3]
    assert sorted(w.L.shard_tile_of_rank(4, 2, r) for r in range(8)) == list(range(8))


def test_rebalancing_moves_the_borders_to_the_bodies(oracle_mod):
    """SURVEY §8(e) "rebalanced every K steps by body count".  A 2 x 2 grid laid out badly for the pile (one column of tiles nearly empty): every
    8 steps the borders move towards equal counts — by what one change may do — and through every switch each body keeps exactly one owner, a
    new owner continues from the old owner's exact state, and the pile behaves."""
    sc = scenes.obb_pile(14, 3, 10, spacing=1.0)
    desc = sharding.tile_grid(sc, 4, 2, 1.5)
    desc.origin_x -= 0.7 * desc.tile_size_x; desc.origin_z += 0.5 * desc.tile_size_z
    ranks = [sharding.ShardedWorld(sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)), desc, r, "local") for r in range(4)]
    s = sc.settings()
    ids = np.arange(sc.num_bodies, dtype=np.uint32)
    first = None; moved = []; handed_over = 0
    truth = copies = owners_before = None
    for i in range(96):
        sharding.step_local(ranks, s, sc.dt)
        owned = [r.world.shard_counts()["owned_bodies"] for r in ranks]
        assert sum(owned) == sc.num_bodies, f"step {i}: {owned}"
        owners = np.full(len(sc.entities), -1)
        for r in ranks:
            e = r.world.shard_owned_entities()
            assert (owners[e] == -1).all(), f"step {i}: a body has two owners"
            owners[e] = r.rank
            if truth is not None:      # whatever a rank owned in this step, it started from the previous owner's exact state — also across a border move
                assert copies[r.rank][e].tobytes() == truth[e].tobytes(), f"step {i} rank {r.rank}: an owner started from a copy that was not current"
        if owners_before is not None and i % 8 == 1:
            handed_over += int((owners != owners_before)[: sc.num_bodies].sum())
        owners_before = owners
        truth = sharding.gather_owned(ranks, sc.num_bodies)    # (asserts the partition once more)
        copies = {r.rank: r.world.get_body_states(ids) for r in ranks}
        if first is None:
            first = owned
        if i % 8 == 7:
            bx, bz = sharding.rebalance_local(ranks)           # in force from step i + 2 (step i + 1 still runs under the old borders and hands over)
            moved.append((float(bx[0]), float(bz[0])))
    assert handed_over > 0.3 * sc.num_bodies, "the border moves handed bodies over"
    assert max(first) > 0.55 * sc.num_bodies and max(owned) < 0.36 * sc.num_bodies, f"{first} -> {owned}"
    assert len(set(moved)) > 3, "the borders moved in several steps (one change is bounded)"
    st = sharding.gather_owned(ranks, sc.num_bodies)
    assert np.isfinite(st).all() and st[:, 1].min() > -0.05


def test_a_rank_never_trusts_a_copy_that_is_not_current(oracle_mod):
    """Why a rank keeps a `known` flag per body.  A body drifts out of the right tile's reach; that rank's last copy of it stays where it left.
    Then the border moves so that the right tile covers that stale position: classified by it, the rank would claim a body that is metres away in
    the left tile.  It must not: the body keeps exactly one owner."""
    parts = [(capi.ENTITY_DYNAMIC, (0.6, 1.0, 0.0), (0, 0, 0, 1), [(capi.SPHERE, (0, 0, 0, 0.3), {})], {"linear_velocity": (-6.0, 0, 0), "gravity_factor": 0.0, "linear_damping": 0.0}),
             (capi.ENTITY_DYNAMIC, (-8.5, 0.5, 0.0), (0, 0, 0, 1), [(capi.AABB, (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5), {})], {}),
             (capi.ENTITY_DYNAMIC, (6.0, 0.5, 0.0), (0, 0, 0, 1), [(capi.AABB, (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5), {})], {}),
             (capi.ENTITY_STATIC, (0, -2.0, 0), (0, 0, 0, 1), [(capi.AABB, (-30, -2, -30, 30, 2, 30), {})], {})]
    sc = scenes.scene_from_parts(parts, iterations=10)
    d = capi.ShardDesc(); d.num_ranks = 2; d.tiles_x = 2; d.tiles_z = 1; d.origin_x = -10.0; d.origin_z = -10.0; d.tile_size_x = 10.0; d.tile_size_z = 20.0; d.ghost_margin = 1.0
    ranks = [sharding.ShardedWorld(sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)), d, r, "local") for r in range(2)]
    s = sc.settings()
    for _ in range(100):                                         # 0.83 s at 6 m/s: the runner is at x = -4.4, far past the right tile's reach (x >= -1)
        sharding.step_local(ranks, s, sc.dt)
    runner = 0
    right_copy = ranks[1].world.get_body_states(np.asarray([runner], np.uint32))[0]
    truth = ranks[0].world.get_body_states(np.asarray([runner], np.uint32))[0]
    assert runner in ranks[0].world.shard_owned_entities() and truth[0] < -4.0 and -1.3 < right_copy[0] < -0.9, "the right rank's copy stopped where the body left its reach"
    for r in ranks:
        r.world.shard_set_borders(np.asarray([-1.6], np.float32), None)      # the right tile now covers the stale copy's position, not the body's
    for i in range(20):
        sharding.step_local(ranks, s, sc.dt)
        owners = [r.rank for r in ranks if runner in r.world.shard_owned_entities()]
        assert owners == [0], f"step {i}: owners of the runner {owners}"
    assert ranks[0].world.shard_get_borders(2, 1)[0][0] == np.float32(-1.6)


def test_gpu_sharding_scenarios_on_the_oracle(oracle_mod):
    """The scenario bodies of tests/test_gpu_sharding.py that only need "a library" run here with the oracle in the product's place: the stale
    copy under a moving border followed by an entity deletion on every rank (what a rank knows follows the bodies through the swap-and-pop), and
    the rebalanced ranks — so the N > 1 logic they cover is exercised on every CPU run as well."""
    import types
    import test_gpu_sharding as G
    lib = types.SimpleNamespace(create_world=lambda device: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    G.test_gpu_rank_never_trusts_a_copy_that_is_not_current(lib)
    G.test_gpu_rebalanced_ranks_match_oracle(lib, oracle_mod, 3, 1, lambda: scenes.ragdolls(8, 2), 3.5)      # islands move whole
    G.test_gpu_cloth_in_a_sharded_world(lib)
    G.test_gpu_sharded_checkpoint_restores_every_ranks_view(lib, oracle_mod)


def test_border_changes_are_validated(oracle_mod):
    sc = _scene()
    w = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    d = sharding._desc_for(sharding.tile_grid(sc, 4, 1, 0.5), 0)
    w.shard_enable(d)
    bx, bz = w.shard_get_borders(4, 1)
    assert len(bx) == 3 and len(bz) == 0 and np.allclose(bx, d.origin_x + d.tile_size_x * np.arange(1, 4))
    m = d.ghost_margin
    for bad in ([bx[1], bx[0], bx[2]],                           # not ascending
                [bx[0], bx[0] + 0.5 * m, bx[2]],                 # a tile narrower than the margin
                [bx[0], bx[2] - 0.5 * m, bx[2] + 0.0],           # border 1 within a margin of old border 2: a body could skip a tile
                [bx[1], bx[1] + 2 * m, bx[2]],                   # border 0 moved onto old border 1
                [np.nan, bx[1], bx[2]]):
        with pytest.raises(capi.PhysicsError):
            w.shard_set_borders(np.asarray(bad, np.float32), None)
    w.shard_set_borders(np.asarray([bx[0] + 0.2, bx[1], bx[2] - 0.2], np.float32), None)
    # the balance arithmetic: equal counts, bounded moves, nothing to do without bodies
    L = w.L
    hist = np.zeros(64, np.uint64); hist[48:] = 10                # everything in the last quarter of [0, 64)
    cur = np.asarray([16.0, 32.0, 48.0], np.float32)
    out = L.shard_balance_borders(hist, 0.0, 64.0, 4, cur, 1.0)
    assert np.array_equal(out, np.asarray([18.0, 34.0, 50.0], np.float32)), out     # one change moves a border by at most two margins (the hand-over must fit one message)
    for _ in range(30):
        out = L.shard_balance_borders(hist, 0.0, 64.0, 4, out, 1.0)
    assert np.array_equal(out, np.asarray([52.0, 56.0, 60.0], np.float32)), out     # ... and the rounds converge to equal counts
    assert np.array_equal(L.shard_balance_borders(np.zeros(64, np.uint64), 0.0, 64.0, 4, cur, 1.0), cur)


def test_global_sweep_axis_does_not_depend_on_the_tiling(oracle_mod):
    """The one global quantity of the pipeline (include/mi_shard.h "Global sweep axis").  The centre statistics are sums of integers, every rank
    sums the colliders it owns (rank 0 also the colliders without a rigid body), and the sums over the ranks are the single world's EXACTLY, as
    9 integers, every step — so every rank sweeps along the single world's axis whatever the tiling, and independent islands (ragdolls that
    touch the ground, not each other) give the single world's result bit for bit under ANY tiling: four x slabs, 2 x 2 tiles."""
    sc = scenes.ragdolls(8, 2, spacing=4.0)
    s = sc.settings(); ids = np.arange(sc.num_bodies, dtype=np.uint32)
    for num_ranks, tiles_z in ((4, 1), (4, 2)):
        ranks, _ = _virtual_ranks(lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL), sc, num_ranks, tiles_z, margin=3.5)
        plain = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
        one = sharding.ShardedWorld(sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)), sharding.tile_grid(sc, 1), 0, "local")   # one tile: the single world, with the shard API
        for i in range(100):
            sharding.step_local(ranks, s, sc.dt); sharding.step_local([one], s, sc.dt); plain.step_fixed(s, sc.dt, 1)
            with np.errstate(over="ignore"):
                total = np.sum([r.world.shard_axis_sums() for r in ranks], axis=0, dtype=np.uint64)
            assert np.array_equal(total, one.world.shard_axis_sums()), f"{num_ranks} ranks, step {i}: the statistic depends on the partition"
            assert {r.world.counts()["sorting_axis"] for r in ranks} == {plain.counts()["sorting_axis"]}, f"step {i}"
            assert sharding.gather_owned(ranks, sc.num_bodies).tobytes() == plain.get_body_states(ids).tobytes(), f"{num_ranks} ranks ({tiles_z} along z), step {i}"
        assert one.world.get_body_states(ids).tobytes() == plain.get_body_states(ids).tobytes()
        assert plain.counts()["num_contacts"] > 100


def test_axis_statistic_known_answers(oracle_mod):
    """ora::axisFromSums / axisTerms (the product's axisFromSums / axisTerms): variance order of quantised centres, exact in 128 bits."""
    import ctypes as C
    f = oracle_mod.library().fn("axis_from_sums", restype=C.c_uint32)

    def axis(points):
        p = np.asarray(points, np.float64)
        q = np.rint(np.clip(p, -1048576.0, 1048576.0) * 1024.0).astype(np.int64)
        s1 = q.sum(axis=0); sq = (q.astype(object) ** 2)
        lo = np.asarray([int(sum(int(v) & 0xFFFFFFFF for v in sq[:, c])) for c in range(3)], dtype=object)
        hi = np.asarray([int(sum(int(v) >> 32 for v in sq[:, c])) for c in range(3)], dtype=object)
        sums = np.asarray([int(v) & 0xFFFFFFFFFFFFFFFF for v in s1] + [int(v) for v in lo] + [int(v) for v in hi], dtype=np.uint64)
        var = [len(p) * (int(hi[c]) * 2 ** 32 + int(lo[c])) - int(s1[c]) ** 2 for c in range(3)]
        expect = (0 if var[0] > var[2] else 2) if var[0] > var[1] else (1 if var[1] > var[2] else 2)
        got = f(sums.ctypes.data_as(C.c_void_p), C.c_uint32(len(p)))
        assert got == expect, (points, var)
        return got
    assert axis([(0, 0, 0), (10, 1, 2)]) == 0
    assert axis([(0, 0, 0), (1, 10, 2)]) == 1
    assert axis([(0, 0, 0), (1, 2, 10)]) == 2
    assert axis([(1, 1, 1), (2, 2, 2)]) == 2                      # all equal: the reference's comparison chain ends at z
    assert axis([(-5e5, 0, 0), (5e5, 1, 1)] * 1000) == 0           # large coordinates, many colliders: no overflow
    assert axis([(3e6, 0, 0), (-3e6, 0.5, 0.25)]) == 0             # clamped to +-2^20 m
    rng = np.random.default_rng(5)
    for _ in range(50):
        axis(rng.normal(0, rng.uniform(0.1, 1e4, 3), (rng.integers(1, 200), 3)))


def test_sharded_checkpoint_restores_every_ranks_view(oracle_mod):
    """A checkpoint of a sharded world is ONE RANK's view: which of its body copies are current, the tile borders in force and the pending ones.
    Save on every rank, run on through a load-balance round and migrations, restore, run again: the same states, bit for bit — and every body
    keeps exactly one owner right after the restore.  (Before the shard section existed, a restore classified with the flags and borders of the
    LATER moment: bodies that had changed rank in between were owned by nobody, or twice, and dropped out of the simulation.)"""
    sc = scenes.obb_pile(16, 3, 8, spacing=1.0)
    desc = _grid(sc, "lopsided", 3, 1)
    ranks = [sharding.ShardedWorld(sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)), desc, r, "local") for r in range(3)]
    s = sc.settings()

    def run(n, start):
        owned = []
        for i in range(start, start + n):
            sharding.step_local(ranks, s, sc.dt)
            owned.append([r.world.shard_counts()["owned_bodies"] for r in ranks])
            assert sum(owned[-1]) == sc.num_bodies, f"step {i}: {owned[-1]}"
            if i % REBALANCE_EVERY == REBALANCE_EVERY - 1:
                sharding.rebalance_local(ranks)
        return owned
    run(12, 0)                                                   # borders have moved once (after step 7, in force from step 9)
    sharding.rebalance_local(ranks)                              # ... and a change is PENDING at the moment of the save
    blobs = [r.world.save_checkpoint() for r in ranks]
    borders_at_save = [r.world.shard_get_borders(3, 1)[0].copy() for r in ranks]
    owned_a = run(28, 12)
    final_a = sharding.gather_owned(ranks, sc.num_bodies)
    assert any(not np.array_equal(r.world.shard_get_borders(3, 1)[0], b) for r, b in zip(ranks, borders_at_save)), "the borders moved on after the save"
    for r, blob in zip(ranks, blobs):
        r.world.load_checkpoint(blob)
    assert all(np.array_equal(r.world.shard_get_borders(3, 1)[0], b) for r, b in zip(ranks, borders_at_save))
    owned_b = run(28, 12)
    assert owned_a == owned_b
    assert sharding.gather_owned(ranks, sc.num_bodies).tobytes() == final_a.tobytes()
    # a blob of another rank, or a rank's blob in an unsharded world, is refused
    with pytest.raises(capi.PhysicsError):
        ranks[0].world.load_checkpoint(blobs[1])
    with pytest.raises(capi.PhysicsError):
        sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)).load_checkpoint(blobs[0])


# ---------------------------------------------------------------------------------------------------------------- exact seam (include/mi_shard.h)
def _exact_case(kind):
    sc = _scene("ragdolls") if kind == "ragdolls" else scenes.obb_pile(12, 4, 8, spacing=1.0)
    return sc, (3.5 if kind == "ragdolls" else 2.5)


@pytest.mark.parametrize("kind,num_ranks,tiles_z", [("pile", 2, 1), ("pile", 3, 1), ("ragdolls", 2, 1), ("pile", 2, 2)], ids=["pile, 2 x slabs", "pile, 3 x slabs", "ragdolls, 2 x slabs", "pile, 2 z slabs"])
def test_exact_seam_virtual_ranks_equal_the_single_world_told_the_tiling(oracle_mod, kind, num_ranks, tiles_z):
    """The exact seam: seam manifolds (all dynamic bodies shared across the same tile border) in the leading colours, solved redundantly by both tiles,
    the owners' velocities of the shared bodies handed over after EVERY sweep.  R ranks then give, bit for bit, what ONE world gives that was told the
    tiling (it only orders its colours the same way) — unlike the default block-Jacobi seam, which differs from any single world after the first step."""
    sc, margin = _exact_case(kind)
    tz = num_ranks if tiles_z == 2 else 1
    desc = sharding.tile_grid(sc, num_ranks, tz, margin)
    make = lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)
    single = sc.populate(make()); single.set_seam_tiling(desc)
    exact = [sharding.ShardedWorld(sc.populate(make()), desc, r, "local") for r in range(num_ranks)]
    jacobi = [sharding.ShardedWorld(sc.populate(make()), desc, r, "local") for r in range(num_ranks)]
    s = sc.settings()
    ents = np.flatnonzero(sc.entities["kind"] != capi.ENTITY_STATIC).astype(np.uint32)
    seam_max = 0
    for i in range(50):
        single.step_fixed(s, sc.dt, 1)
        sharding.step_local_exact(exact, s, sc.dt); sharding.step_local(jacobi, s, sc.dt)
        assert sharding.gather_owned(exact, len(ents)).tobytes() == single.get_body_states(ents).tobytes(), f"step {i}"
        assert sum(r.world.shard_counts()["owned_contacts"] for r in exact) == single.counts()["num_contacts"]
        st = single.seam_stats(); seam_max = max(seam_max, st["seam_manifolds"])
        assert st["violations"] == 0 and all(r.world.seam_stats()["violations"] == 0 for r in exact)
        assert st["seam_manifolds"] == max(r.world.seam_stats()["seam_manifolds"] for r in exact) or num_ranks > 2
    assert seam_max > 0, "no manifold ever lay on the seam"
    if kind == "pile":
        assert sharding.gather_owned(jacobi, len(ents)).tobytes() != single.get_body_states(ents).tobytes(), "block Jacobi is not expected to equal a single world"


def _exact_worker(rank, world_size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, str(ROOT))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    import oracle
    sc, margin = _exact_case("pile")
    desc = sharding.tile_grid(sc, world_size, 1, margin)
    sw = sharding.ShardedWorld(sc.populate(oracle.create_world(oracle.ORDER_CANONICAL)), desc, rank, "dist", dist)
    sw.enable_exact_seam()
    s = sc.settings()
    for _ in range(30):
        sw.step(s, sc.dt)
    ents, st = sw.owned_states()
    np.savez(Path(out_dir) / f"rank{rank}.npz", ents=ents, states=st, violations=sw.world.seam_stats()["violations"])
    dist.barrier(); dist.destroy_process_group()


def test_exact_seam_processes_over_gloo_equal_the_single_world(tmp_path, oracle_mod):
    """Two real processes, the per-sweep messages over gloo from inside the library's sweep callback: the union of what the ranks own is the single
    world told the tiling, bit for bit."""
    port = 29900 + (os.getpid() % 2000)
    mp.spawn(_exact_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sc, margin = _exact_case("pile")
    desc = sharding.tile_grid(sc, 2, 1, margin)
    single = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)); single.set_seam_tiling(desc)
    s = sc.settings()
    for _ in range(30):
        single.step_fixed(s, sc.dt, 1)
    seen = {}
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert int(got["violations"]) == 0
        for e, st in zip(got["ents"], got["states"]):
            assert int(e) not in seen; seen[int(e)] = st
    ents = np.asarray(sorted(seen), np.uint32)
    assert len(ents) == sc.num_bodies
    assert np.stack([seen[int(e)] for e in ents]).tobytes() == single.get_body_states(ents).tobytes()


def test_exact_seam_conditions_are_checked(oracle_mod):
    """Slabs only; tiles at least two margins wide; a margin that does not cover the reach of a contact is REPORTED (violations), never silent."""
    sc, _ = _exact_case("pile")
    make = lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)
    w = sc.populate(make()); w.shard_enable(sharding._desc_for(sharding.tile_grid(sc, 4, 2, 2.5), 0))
    with pytest.raises(capi.PhysicsError):
        w.shard_set_exact_seam(True, None)                       # 2 x 2 tiles: a corner body is seen by four tiles
    narrow = sharding.tile_grid(sc, 2, 1, 2.5); narrow.ghost_margin = 0.6 * narrow.tile_size_x
    w2 = sc.populate(make())
    with pytest.raises(capi.PhysicsError):
        w2.set_seam_tiling(narrow)
    desc = sharding.tile_grid(sc, 2, 1, 0.2)                     # boxes are up to 1.2 m across: a 0.2 m margin cannot cover a contact
    ranks = [sharding.ShardedWorld(sc.populate(make()), desc, r, "local") for r in range(2)]
    s = sc.settings()
    for _ in range(40):
        sharding.step_local_exact(ranks, s, sc.dt)
    assert sum(r.world.seam_stats()["violations"] for r in ranks) > 0


def test_exact_seam_checkpoint_round_trip(oracle_mod):
    """Save every rank in the middle of an exact-seam run, go on, restore (the mode is set on the worlds — it is not part of the blob), go on again:
    the same states, bit for bit, and still the single world told the tiling."""
    sc, margin = _exact_case("pile")
    desc = sharding.tile_grid(sc, 2, 1, margin)
    make = lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)
    ranks = [sharding.ShardedWorld(sc.populate(make()), desc, r, "local") for r in range(2)]
    single = sc.populate(make()); single.set_seam_tiling(desc)
    s = sc.settings()
    ents = np.flatnonzero(sc.entities["kind"] != capi.ENTITY_STATIC).astype(np.uint32)
    for _ in range(15):
        sharding.step_local_exact(ranks, s, sc.dt); single.step_fixed(s, sc.dt, 1)
    blobs = [r.world.save_checkpoint() for r in ranks]
    for _ in range(15):
        sharding.step_local_exact(ranks, s, sc.dt); single.step_fixed(s, sc.dt, 1)
    final = sharding.gather_owned(ranks, len(ents))
    assert final.tobytes() == single.get_body_states(ents).tobytes()
    for r, blob in zip(ranks, blobs):
        r.world.load_checkpoint(blob)
    for _ in range(15):
        sharding.step_local_exact(ranks, s, sc.dt)
    assert sharding.gather_owned(ranks, len(ents)).tobytes() == final.tobytes()
