"""Sharded worlds on the GPU (include/mi_shard.h, csrc k_shard_*): several ranks of one scene as several worlds of ONE process on one
GPU, the neighbour messages moved through the caller's-transport entry points (mi_world_shard_export / _import).  The CPU oracle
mirrors the sharding, so every rank is compared with its oracle twin bit for bit; tests/test_distributed.py shows that real
processes over a real transport give exactly what such virtual ranks give."""
import numpy as np
import pytest

from d3d12renderer_amd import capi, scenes, sharding

pytestmark = pytest.mark.gpu


def _ranks(make_world, sc, n, tiles_z=1, margin=2.5):
    desc = sharding.tile_grid(sc, n, tiles_z, margin)
    return [sharding.ShardedWorld(sc.populate(make_world()), desc, r, "local") for r in range(n)]


@pytest.mark.parametrize("n,tiles_z,make,margin", [(3, 1, lambda: scenes.obb_pile(12, 4, 8, spacing=1.0), 2.5), (4, 2, lambda: scenes.mixed_stack(10, 4, 10), 2.5),
                                                   (2, 1, lambda: scenes.shape_zoo(), 2.5), (3, 1, lambda: scenes.ragdolls(6, 3), 3.5), (2, 2, lambda: scenes.vehicles(4, 4), 6.0),
                                                   (2, 2, lambda: scenes.terrain_field(9, 2, 9, with_unsupported=False), 2.5)],
                         ids=["3 slabs boxes", "2x2 tiles mixed", "2 slabs all shapes", "3 slabs ragdolls (cfg4)", "2x1 tiles vehicles (cfg5)", "2x2 tiles on heightmap terrain"])
def test_gpu_virtual_ranks_match_oracle_virtual_ranks(mi_lib, oracle_mod, n, tiles_z, make, margin):
    """Joint scenes too: an articulated island (ragdoll, vehicle) is owned / ghosted / ignored as one, decided by its root body."""
    sc = make()
    g = _ranks(lambda: mi_lib.create_world(0), sc, n, tiles_z, margin)
    o = _ranks(lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL), sc, n, tiles_z, margin)
    s = sc.settings()
    migrated = False; first = None
    for i in range(100):
        sharding.step_local(g, s, sc.dt); sharding.step_local(o, s, sc.dt)
        for a, b in zip(g, o):
            assert a.world.counts() == b.world.counts(), f"step {i} rank {a.rank}: local counts"
            assert a.world.shard_counts() == b.world.shard_counts(), f"step {i} rank {a.rank}: owned counts"
        if i % 10 == 0 or i == 99:
            for a, b in zip(g, o):
                ea, sa = a.owned_states(); eb, sb = b.owned_states()
                assert np.array_equal(ea, eb) and sa.tobytes() == sb.tobytes(), f"step {i} rank {a.rank}: owned states"
            owners = np.concatenate([np.full(len(r.world.shard_owned_entities()), r.rank) for r in g])
            ents = np.concatenate([r.world.shard_owned_entities() for r in g])
            assert len(np.unique(ents)) == sc.num_bodies == len(ents)
            cur = owners[np.argsort(ents)]
            if first is None:
                first = cur
            migrated |= bool((cur != first).any())
    assert migrated or n != 3 or margin != 2.5, "no body changed owner in the spreading box pile"


def _lopsided(sc, n, tiles_z, margin):
    """A tile grid laid out badly for the scene: one column of tiles nearly empty, so the load balance has work to do."""
    desc = sharding.tile_grid(sc, n, tiles_z, margin)
    desc.origin_x -= 0.7 * desc.tile_size_x
    if tiles_z > 1:
        desc.origin_z += 0.5 * desc.tile_size_z
    return desc


@pytest.mark.parametrize("n,tiles_z,make,margin", [(3, 1, lambda: scenes.obb_pile(16, 3, 8, spacing=1.0), 1.5), (4, 2, lambda: scenes.mixed_stack(12, 3, 10), 1.5),
                                                   (3, 1, lambda: scenes.ragdolls(8, 2), 3.5)],
                         ids=["3 slabs boxes", "2x2 tiles mixed", "3 slabs ragdolls"])
def test_gpu_rebalanced_ranks_match_oracle(mi_lib, oracle_mod, n, tiles_z, make, margin):
    """Load balance (mi_world_shard_histogram / mi_shard_balance_borders / mi_world_shard_set_borders): the borders move every 8 steps, on the
    GPU ranks and on their oracle twins — same histograms, same borders, and every count and owned state stays bit-identical through the hand-overs."""
    sc = make()
    desc = _lopsided(sc, n, tiles_z, margin)
    g = [sharding.ShardedWorld(sc.populate(mi_lib.create_world(0)), desc, r, "local") for r in range(n)]
    o = [sharding.ShardedWorld(sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)), desc, r, "local") for r in range(n)]
    s = sc.settings()
    first = None; borders = [np.concatenate(g[0].world.shard_get_borders(desc.tiles_x, desc.tiles_z)).tobytes()]
    for i in range(80):
        sharding.step_local(g, s, sc.dt); sharding.step_local(o, s, sc.dt)
        owned = [a.world.shard_counts()["owned_bodies"] for a in g]
        assert sum(owned) == sc.num_bodies
        first = first or owned
        for a, b in zip(g, o):
            assert a.world.counts() == b.world.counts(), f"step {i} rank {a.rank}: local counts"
            assert a.world.shard_counts() == b.world.shard_counts(), f"step {i} rank {a.rank}: owned counts"
        if i % 8 == 7 or i == 79:
            for a, b in zip(g, o):
                ea, sa = a.owned_states(); eb, sb = b.owned_states()
                assert np.array_equal(ea, eb) and sa.tobytes() == sb.tobytes(), f"step {i} rank {a.rank}: owned states"
                assert np.array_equal(a.histograms(), b.histograms()), f"step {i} rank {a.rank}: histograms"
            bg = sharding.rebalance_local(g); bo = sharding.rebalance_local(o)
            assert bg[0].tobytes() == bo[0].tobytes() and bg[1].tobytes() == bo[1].tobytes()
            borders.append(bg[0].tobytes() + bg[1].tobytes())
    assert len(set(borders)) > 1, "the borders moved"
    assert max(owned) < max(first) - 0.1 * sc.num_bodies, f"load balance: {first} -> {owned}"


@pytest.mark.parametrize("n,tiles_z,make,margin", [(4, 1, lambda: scenes.obb_pile(128, 4, 16, spacing=1.0), 1.5), (4, 2, lambda: scenes.obb_pile(48, 4, 48, spacing=1.0), 1.5)],
                         ids=["4 slabs, 8 192 boxes", "2x2 tiles, 9 216 boxes"])
def test_gpu_block_skipping_changes_nothing(mi_lib, monkeypatch, n, tiles_z, make, margin):
    """A rank's per-body and per-collider passes visit only the blocks of 256 with something simulated in them (kernels.hpp, shardBlockRecent: classification and
    packing look at the blocks that were active or received a record within the last two steps, the integrators and the collider pass at those with a body simulated
    now or in the previous step).  Ranks that skip against ranks that visit every block (MI_SHARD_BLOCK_SKIP=0), over a badly laid out grid whose borders move every
    8 steps — bodies change owner by the hundred, blocks fall dead and come back: every count, every owned state and every border bit-identical, every step."""
    sc = make()
    desc = _lopsided(sc, n, tiles_z, margin)
    worlds = {}
    for skip in ("1", "0"):
        monkeypatch.setenv("MI_SHARD_BLOCK_SKIP", skip)
        worlds[skip] = [sharding.ShardedWorld(sc.populate(mi_lib.create_world(0)), desc, r, "local") for r in range(n)]
    a_, b_ = worlds["1"], worlds["0"]
    s = sc.settings()
    first = None
    for i in range(72):
        sharding.step_local(a_, s, sc.dt); sharding.step_local(b_, s, sc.dt)
        owned = [a.world.shard_counts()["owned_bodies"] for a in a_]
        assert sum(owned) == sc.num_bodies
        first = first or owned
        for a, b in zip(a_, b_):
            assert a.world.counts() == b.world.counts(), f"step {i} rank {a.rank}: local counts"
            assert a.world.shard_counts() == b.world.shard_counts(), f"step {i} rank {a.rank}: owned counts"
            ea, sa = a.owned_states(); eb, sb = b.owned_states()
            assert np.array_equal(ea, eb) and sa.tobytes() == sb.tobytes(), f"step {i} rank {a.rank}: owned states"
        if i % 8 == 7:
            ba = sharding.rebalance_local(a_); bb = sharding.rebalance_local(b_)
            assert ba[0].tobytes() == bb[0].tobytes() and ba[1].tobytes() == bb[1].tobytes()
    assert max(owned) < max(first), f"load balance moved bodies between the ranks: {first} -> {owned}"


def test_gpu_sweep_axis_follows_the_exchange_on_ranks_that_skip_collider_block_0(mi_lib, monkeypatch):
    """The global sweep axis changes while three of four ranks simulate nothing in collider block 0 (round-5 advisor item: the step's axis word was written by the lane of
    collider 0, inside the visit of a block a sharded rank may skip — such a rank kept sweeping along the old axis and reported it in counts().sorting_axis).  A long
    line of boxes along x (last created = lowest collider indices = the right-most slab) and two flights of boxes closing in along z: the variance along z falls below
    the one along x after a dozen steps.  Ranks that skip blocks == ranks that visit every block, every step, and every rank reports the same axis."""
    box = [(capi.AABB, (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5), {})]
    parts = [(capi.ENTITY_STATIC, (0, -2.0, 0), (0, 0, 0, 1), [(capi.AABB, (-120, -2, -250, 120, 2, 250), {})], {})]
    for k in range(320):     # the flights (created first: highest collider indices), over the left-most slab, no gravity
        side = 1.0 if k % 2 else -1.0
        parts.append((capi.ENTITY_DYNAMIC, (-80.0 + 0.11 * k, 6.0 + 1.2 * (k % 5), side * (150.0 + (k % 7))), (0, 0, 0, 1), box,
                      {"linear_velocity": (0.0, 0.0, -side * 240.0), "gravity_factor": 0.0, "linear_damping": 0.0}))
    for ix in range(160):    # the line, ascending x
        for iz in range(8):
            parts.append((capi.ENTITY_DYNAMIC, (-83.475 + 1.05 * ix, 0.5, -3.675 + 1.05 * iz), (0, 0, 0, 1), box, {}))
    sc = scenes.scene_from_parts(parts, iterations=6)
    d = capi.ShardDesc(); d.num_ranks = 4; d.tiles_x = 4; d.tiles_z = 1; d.origin_x = -84.0; d.origin_z = -400.0; d.tile_size_x = 42.0; d.tile_size_z = 800.0; d.ghost_margin = 1.5
    worlds = {}
    for skip in ("1", "0"):
        monkeypatch.setenv("MI_SHARD_BLOCK_SKIP", skip)
        worlds[skip] = [sharding.ShardedWorld(sc.populate(mi_lib.create_world(0)), d, r, "local") for r in range(4)]
    a_, b_ = worlds["1"], worlds["0"]
    s = sc.settings()
    axes = []
    for i in range(40):
        sharding.step_local(a_, s, sc.dt); sharding.step_local(b_, s, sc.dt)
        per_rank = [a.world.counts()["sorting_axis"] for a in a_]
        assert len(set(per_rank)) == 1, f"step {i}: the ranks sweep along different axes: {per_rank}"
        axes.append(per_rank[0])
        for a, b in zip(a_, b_):
            assert a.world.counts() == b.world.counts(), f"step {i} rank {a.rank}: local counts"
            assert a.world.shard_counts() == b.world.shard_counts(), f"step {i} rank {a.rank}: owned counts"
            ea, sa = a.owned_states(); eb, sb = b.owned_states()
            assert np.array_equal(ea, eb) and sa.tobytes() == sb.tobytes(), f"step {i} rank {a.rank}: owned states"
    assert len(set(axes)) > 1, f"the sweep axis never changed: {sorted(set(axes))}"


def test_gpu_rank_never_trusts_a_copy_that_is_not_current(mi_lib):
    """tests/test_distributed.py::test_a_rank_never_trusts_a_copy_that_is_not_current on the GPU: a border moves over the place where a rank last
    saw a body that has long left — the rank must not claim it.  Then entities are deleted on every rank (re-upload of everything, body indices
    shift): what a rank knows follows the bodies."""
    parts = [(capi.ENTITY_DYNAMIC, (0.6, 1.0, 0.0), (0, 0, 0, 1), [(capi.SPHERE, (0, 0, 0, 0.3), {})], {"linear_velocity": (-6.0, 0, 0), "gravity_factor": 0.0, "linear_damping": 0.0}),
             (capi.ENTITY_DYNAMIC, (-8.5, 0.5, 0.0), (0, 0, 0, 1), [(capi.AABB, (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5), {})], {}),
             (capi.ENTITY_DYNAMIC, (6.0, 0.5, 0.0), (0, 0, 0, 1), [(capi.AABB, (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5), {})], {}),
             (capi.ENTITY_DYNAMIC, (7.5, 0.5, 0.0), (0, 0, 0, 1), [(capi.AABB, (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5), {})], {}),
             (capi.ENTITY_STATIC, (0, -2.0, 0), (0, 0, 0, 1), [(capi.AABB, (-30, -2, -30, 30, 2, 30), {})], {})]
    sc = scenes.scene_from_parts(parts, iterations=10)
    d = capi.ShardDesc(); d.num_ranks = 2; d.tiles_x = 2; d.tiles_z = 1; d.origin_x = -10.0; d.origin_z = -10.0; d.tile_size_x = 10.0; d.tile_size_z = 20.0; d.ghost_margin = 1.0
    ranks = [sharding.ShardedWorld(sc.populate(mi_lib.create_world(0)), d, r, "local") for r in range(2)]
    s = sc.settings()
    for _ in range(100):
        sharding.step_local(ranks, s, sc.dt)
    right_copy = ranks[1].world.get_body_states(np.asarray([0], np.uint32))[0]
    truth = ranks[0].world.get_body_states(np.asarray([0], np.uint32))[0]
    assert 0 in ranks[0].world.shard_owned_entities() and truth[0] < -4.0 and -1.3 < right_copy[0] < -0.9
    for r in ranks:
        r.world.shard_set_borders(np.asarray([-1.6], np.float32), None)
    for i in range(6):
        sharding.step_local(ranks, s, sc.dt)
        assert [r.rank for r in ranks if 0 in r.world.shard_owned_entities()] == [0], f"step {i}"
    for r in ranks:                                              # entity 2 (a body of the right tile) goes: the runner stays body 0, body 3 moves into slot 2 (swap and pop)
        r.world.destroy_entity(2)
    for i in range(6):
        sharding.step_local(ranks, s, sc.dt)
        owned = [sorted(r.world.shard_owned_entities().tolist()) for r in ranks]
        assert owned == [[0, 1], [3]], f"step {i} after the deletion: {owned}"


def test_gpu_cloth_in_a_sharded_world(mi_lib):
    """Cloths do not interact with rigid bodies (cloth.cpp): every rank steps all of them, identically — the same vertices on every
    rank and in the unsharded world, while the rigid bodies are sharded as usual."""
    sc = scenes.obb_pile(8, 3, 8, spacing=1.0)
    worlds = [sc.populate(mi_lib.create_world(0)) for _ in range(3)]
    for w in worlds:
        c = w.create_cloth(3.0, 2.0, 16, 12, 4.0, stiffness=0.6, damping=0.4)
        w.set_cloth_fixed_vertices(c, (0.0, 6.0, 0.0), move_rigid=True)
    desc = sharding.tile_grid(sc, 2)
    ranks = [sharding.ShardedWorld(worlds[r], desc, r, "local") for r in range(2)]
    s = sc.settings()
    for _ in range(60):
        sharding.step_local(ranks, s, sc.dt); worlds[2].step_fixed(s, sc.dt, 1)
    ref = worlds[2].cloth_state(0, 16 * 12)
    for r in ranks:
        got = r.world.cloth_state(0, 16 * 12)
        assert np.isfinite(got[0]).all() and got[0].tobytes() == ref[0].tobytes() and got[1].tobytes() == ref[1].tobytes()
    assert sum(r.world.shard_counts()["owned_bodies"] for r in ranks) == sc.num_bodies


def test_gpu_one_tile_is_the_unsharded_world(mi_lib):
    """Sharding with a single tile owns everything: the activity mask, the dead-collider path and the owner-only integration must
    then be invisible — bit-identical to the same world without sharding."""
    sc = scenes.obb_pile(10, 4, 10, spacing=1.0)
    a = sc.populate(mi_lib.create_world(0)); b = sc.populate(mi_lib.create_world(0))
    sw = sharding.ShardedWorld(b, sharding.tile_grid(sc, 1), 0, "local")
    s = sc.settings()
    for i in range(80):
        a.step_fixed(s, sc.dt, 1); sw.step(s, sc.dt)
        assert a.counts() == b.counts(), f"step {i}"
    assert a.physics_transforms()[0].tobytes() == b.physics_transforms()[0].tobytes()
    assert b.shard_counts()["owned_bodies"] == sc.num_bodies and b.shard_counts()["owned_contacts"] == b.counts()["num_contacts"]


def test_gpu_bench_scene_in_two_tiles(mi_lib):
    """The 262 144-body pile cut into two x tiles (strong scaling, both ranks on this one GPU): each rank simulates its half plus
    the ghost strip, the owned sets partition the bodies, owned contacts add up to a settled pile's, and each rank's step is
    cheaper than the whole world's."""
    sc = scenes.obb_pile(128, 16, 128)
    ranks = _ranks(lambda: mi_lib.create_world(0), sc, 2)
    s = sc.settings()
    for _ in range(160):
        sharding.step_local(ranks, s, sc.dt)
    owned = [r.world.shard_counts() for r in ranks]
    assert sum(o["owned_bodies"] for o in owned) == sc.num_bodies
    assert abs(owned[0]["owned_bodies"] - owned[1]["owned_bodies"]) < 0.1 * sc.num_bodies
    assert sum(o["owned_contacts"] for o in owned) > 1.0 * sc.num_bodies
    for r in ranks:                                   # a rank sees its half + ghosts, not the whole pile
        assert r.world.counts()["num_contacts"] < 0.75 * sum(o["owned_contacts"] for o in owned)
    st = np.concatenate([r.owned_states()[1] for r in ranks])
    assert np.isfinite(st).all() and st[:, 1].min() > -0.05


def test_gpu_library_rccl_transport_comes_up(mi_lib):
    """The library's own transport with ONE rank on this one GPU: librccl is found at run time (dlopen), ncclGetUniqueId and
    ncclCommInitRank succeed, a sharded step runs its (empty) send / receive group on the world's stream and the communicator is
    destroyed with the world.  (Sends and receives themselves need one GPU per rank; RCCL refuses two ranks on one device.)"""
    sc = scenes.obb_pile(8, 4, 8, spacing=1.0)
    plain = sc.populate(mi_lib.create_world(0))
    w = sc.populate(mi_lib.create_world(0))
    desc = sharding._desc_for(sharding.tile_grid(sc, 1), 0)
    w.shard_enable(desc)
    ident = w.L.shard_unique_id()
    assert len(ident) == 128 and any(ident)
    w.shard_attach_rccl(ident)
    s = sc.settings()
    w.step_fixed(s, sc.dt, 30); plain.step_fixed(s, sc.dt, 30)       # with the library transport several internal steps per call are fine
    assert w.physics_transforms()[0].tobytes() == plain.physics_transforms()[0].tobytes()
    assert w.shard_counts()["owned_bodies"] == sc.num_bodies
    assert np.array_equal(w.shard_allreduce_u64([3, 1 << 40, 0]), np.asarray([3, 1 << 40, 0], np.uint64))   # one rank: the sum is the value (ncclAllReduce resolved and run)
    w.shard_rebalance(64)                                             # a whole round through the library (one tile: no border to move)
    w.shard_detach_rccl()                                             # back to the caller's transport (what a rank does when a peer could not attach)
    w.step_fixed(s, sc.dt, 1); plain.step_fixed(s, sc.dt, 1)
    assert w.physics_transforms()[0].tobytes() == plain.physics_transforms()[0].tobytes()
    with pytest.raises(capi.PhysicsError):
        w.shard_rebalance(64)                                         # caller's transport: the caller reduces the histograms
    w.close()


@pytest.mark.parametrize("exact", [False, True], ids=["block-Jacobi seam", "exact seam: one hand-over per sweep"])
def test_gpu_library_transport_loops_back_on_one_gpu(mi_lib, exact):
    """The library's RCCL path with REAL records on one GPU: rank 0 of a two-tile grid on a one-rank communicator whose neighbour is the
    rank itself (mi_debug_shard_attach_loopback).  Every exchange runs pack -> ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd ->
    k_shard_unpack -> ncclAllReduce on the world's stream (with the exact seam also the per-sweep k_seam_sweep_pack -> send / receive ->
    k_seam_sweep_unpack), and what arrives is, bit for bit, what was sent.  A second world on the CALLER's transport is handed its own
    messages back by this test; the two must stay bit-identical step after step: the transports are interchangeable."""
    sc = scenes.obb_pile(16, 4, 8, spacing=1.0)
    desc = sharding._desc_for(sharding.tile_grid(sc, 2, 1, 2.5), 0)
    a = sc.populate(mi_lib.create_world(0)); b = sc.populate(mi_lib.create_world(0))
    a.shard_enable(desc); b.shard_enable(desc)
    a.shard_attach_loopback()
    assert a.shard_neighbours() == [0], "rank 0 of two x tiles has one neighbour; looped back, it is rank 0 itself"
    sweeps_seen = []
    if exact:
        a.shard_set_exact_seam(True, None)
        def hand_back(sweep):
            sweeps_seen.append(sweep)
            b.shard_import_sweep(b.shard_export_sweep(0))
        b.shard_set_exact_seam(True, hand_back)
    s = sc.settings()
    moved = 0; prev_n = last_n = 0
    for i in range(60):
        a.step_fixed(s, sc.dt, 1)
        b.step_fixed(s, sc.dt, 1)
        msg = b.shard_export(0); b.shard_import(msg); b.shard_set_axis_sums(b.shard_axis_sums())
        sent = a.shard_export(0); got = a.shard_peek_received(0)
        n = int(sent[:1].view(np.uint32)[0])
        assert n == int(msg[:1].view(np.uint32)[0])
        used = (n + 1) * (len(sent) // (desc.max_records + 1))
        assert sent[:used].tobytes() == got[:used].tobytes(), f"step {i}: what RCCL delivered is not what was packed"
        per = len(sent) // (desc.max_records + 1)
        recs13 = lambda m: np.sort(m[per:used].reshape(-1, per).view(np.uint32).view([("f%d" % k, np.uint32) for k in range(per)]).ravel())
        assert recs13(sent).tobytes() == recs13(msg).tobytes(), f"step {i}: records of the two transports (appended to by wave-aggregated atomics: compared as sets)"
        moved += n; prev_n, last_n = last_n, n
        assert a.counts() == b.counts(), f"step {i}"
        pa, qa = a.physics_transforms(); pb, qb = b.physics_transforms()
        assert pa.tobytes() == pb.tobytes() and qa.tobytes() == qb.tobytes(), f"step {i}"
        if exact:   # the last sweep's hand-over: what came back is what was packed (the list of shared bodies is appended to by atomics: its ORDER may differ from the other world's)
            sw = a.shard_peek_received(0, sweep_message=True); own = a.shard_export_sweep(0)
            ns = int(sw[:1].view(np.uint32)[0])
            assert ns > 0 and sw[:8 * (ns + 1)].tobytes() == own.tobytes(), f"step {i}: sweep message"
            other = b.shard_export_sweep(0)
            recs = lambda m: np.sort(m[8:].reshape(-1, 8).view(np.uint32).view([("f%d" % k, np.uint32) for k in range(8)]).ravel())
            assert len(other) == len(own) and recs(own).tobytes() == recs(other).tobytes(), f"step {i}: sweep records of the two transports"
    assert moved > 60, "the seam strip holds bodies: records must have travelled"
    if exact:
        assert len(sweeps_seen) == 60 * s.num_rigid_solver_iterations
    st = a.shard_exchange_stats()
    assert st["exchanges"] == 60
    # the messages travel as long as the previous exchange made them (either direction) x 1.5 + 512, full size only right after attach: both ends derive the same number
    assert st["message_records_last"] == [min(desc.max_records, prev_n + prev_n // 2 + 512)], (st["message_records_last"], prev_n, desc.max_records)
    assert st["message_records_last"][0] < desc.max_records
    if exact:
        assert 0 < st["sweep_message_bytes"] < (desc.max_records + 1) * 32, "the sweep messages are sized from the previous step's lists as well"
    assert st["message_bytes_sum"] <= 60 * (desc.max_records + 1) * 56
    a.close(); b.close()


def test_gpu_global_sweep_axis_and_independent_islands_under_any_tiling(mi_lib, oracle_mod):
    """include/mi_shard.h "Global sweep axis": the 9 integer centre statistics summed over the ranks equal the single world's exactly, so all
    ranks sweep along the single world's axis and independent islands (cfg4: ragdolls on the ground; cfg5: vehicles on hull tiles, whose
    gear-tooth and wheel pairs are of EQUAL shape type — the pairs the axis orients) give the single world's result bit for bit, whatever the
    tiling.  GPU ranks, the GPU single world and the oracle's ranks all agree."""
    for sc, grids, margin, steps in ((scenes.ragdolls(8, 2, spacing=4.0), ((4, 1), (4, 2)), 3.5, 80), (scenes.vehicles(4, 2), ((2, 1), (4, 2)), 6.0, 60)):
        s = sc.settings(); ids = np.arange(sc.num_bodies, dtype=np.uint32)
        for num_ranks, tiles_z in grids:
            g = _ranks(lambda: mi_lib.create_world(0), sc, num_ranks, tiles_z, margin)
            o = _ranks(lambda: oracle_mod.create_world(oracle_mod.ORDER_CANONICAL), sc, num_ranks, tiles_z, margin)
            plain = sc.populate(mi_lib.create_world(0))
            one = sharding.ShardedWorld(sc.populate(mi_lib.create_world(0)), sharding.tile_grid(sc, 1), 0, "local")
            for i in range(steps):
                sharding.step_local(g, s, sc.dt); sharding.step_local(o, s, sc.dt); sharding.step_local([one], s, sc.dt); plain.step_fixed(s, sc.dt, 1)
                with np.errstate(over="ignore"):
                    total = np.sum([r.world.shard_axis_sums() for r in g], axis=0, dtype=np.uint64)
                assert np.array_equal(total, one.world.shard_axis_sums()), f"{num_ranks} ranks, step {i}: the statistic depends on the partition"
                for a, b in zip(g, o):
                    assert np.array_equal(a.world.shard_axis_sums(), b.world.shard_axis_sums()), f"step {i} rank {a.rank}: GPU sums != oracle sums"
                assert {r.world.counts()["sorting_axis"] for r in g} == {plain.counts()["sorting_axis"]}, f"step {i}"
                if i % 10 == 9 or i == steps - 1:
                    assert sharding.gather_owned(g, sc.num_bodies).tobytes() == plain.get_body_states(ids).tobytes(), f"{num_ranks} ranks ({tiles_z} along z), step {i}: sharded != single world"
            assert plain.counts()["num_contacts"] > 50
            for r in g + [one]:
                r.world.close()
            plain.close()


def test_gpu_sharded_checkpoint_restores_every_ranks_view(mi_lib, oracle_mod):
    """mi_world_save_checkpoint / _load_checkpoint on the ranks of a sharded world (shard section of the blob: current copies, borders in force
    and pending): save mid-run with a border change pending, run on through migrations, restore, run again — bit-identical; the oracle's ranks
    load the GPU ranks' blobs and continue identically; another rank's blob is refused."""
    sc = scenes.obb_pile(16, 3, 8, spacing=1.0)
    desc = _lopsided(sc, 3, 1, 1.5)
    g = [sharding.ShardedWorld(sc.populate(mi_lib.create_world(0)), desc, r, "local") for r in range(3)]
    s = sc.settings()

    def run(ranks, n, start):
        owned = []
        for i in range(start, start + n):
            sharding.step_local(ranks, s, sc.dt)
            owned.append([r.world.shard_counts()["owned_bodies"] for r in ranks])
            assert sum(owned[-1]) == sc.num_bodies, f"step {i}: {owned[-1]}"
            if i % 8 == 7:
                sharding.rebalance_local(ranks)
        return owned
    run(g, 12, 0)
    sharding.rebalance_local(g)                                   # a change is pending at the moment of the save
    blobs = [r.world.save_checkpoint() for r in g]
    borders_at_save = [r.world.shard_get_borders(3, 1)[0].copy() for r in g]
    owned_a = run(g, 28, 12)
    final_a = sharding.gather_owned(g, sc.num_bodies)
    assert any(not np.array_equal(r.world.shard_get_borders(3, 1)[0], b) for r, b in zip(g, borders_at_save))
    for r, blob in zip(g, blobs):
        r.world.load_checkpoint(blob)
    assert all(np.array_equal(r.world.shard_get_borders(3, 1)[0], b) for r, b in zip(g, borders_at_save))
    assert run(g, 28, 12) == owned_a
    assert sharding.gather_owned(g, sc.num_bodies).tobytes() == final_a.tobytes()
    o = [sharding.ShardedWorld(sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)), desc, r, "local") for r in range(3)]
    for r, blob in zip(o, blobs):
        r.world.load_checkpoint(blob)
    assert run(o, 28, 12) == owned_a
    assert sharding.gather_owned(o, sc.num_bodies).tobytes() == final_a.tobytes()
    with pytest.raises(capi.PhysicsError):
        g[0].world.load_checkpoint(blobs[1])
    with pytest.raises(capi.PhysicsError):
        sc.populate(mi_lib.create_world(0)).load_checkpoint(blobs[0])


def test_gpu_exact_seam_guard_rails(mi_lib):
    """What the exact seam needs is checked where it is set up, not found out by a drifting simulation: a hand-over after every sweep (a transport or a callback),
    tiles at least two ghost margins wide — when the mode is switched on AND whenever the borders move afterwards (a body within the margin of two borders has no
    single seam class) —, and `ShardedWorld.check_seam` turns a violated seam class into an error."""
    sc = scenes.obb_pile(24, 3, 6, spacing=1.0)
    desc = sharding.tile_grid(sc, 4, 1, 1.5)                     # four x-slabs, 6 m each, margin 1.5 m
    w = sc.populate(mi_lib.create_world(0))
    sw = sharding.ShardedWorld(w, desc, 1, "local")
    moves = lambda sweep: 0
    with pytest.raises(capi.PhysicsError):
        w.shard_set_exact_seam(True, None)                       # caller's transport and nobody to call after a sweep
    bx, _ = w.shard_get_borders(desc.tiles_x, desc.tiles_z)
    narrow = np.array([bx[0], bx[0] + 2.0, bx[2]], np.float32)   # tile 1 two metres wide: more than one margin (fine for block Jacobi), less than two
    w.shard_set_borders(narrow, None)
    s = sc.settings()
    w.step_fixed(s, sc.dt, 1)                                    # (the step's exchange puts them in force)
    with pytest.raises(capi.PhysicsError):
        w.shard_set_exact_seam(True, moves)                      # borders in force (or pending) too close for the exact seam
    w.shard_set_borders(bx, None)
    w.step_fixed(s, sc.dt, 1)
    w.shard_set_exact_seam(True, moves)
    with pytest.raises(capi.PhysicsError):
        w.shard_set_borders(narrow, None)                        # ... and they cannot get that close afterwards
    sw.exact = True
    assert sw.check_seam()["violations"] == 0
    w.shard_set_exact_seam(False, None)


def test_gpu_shard_entry_points_reject_misuse(mi_lib):
    """mi_world_shard_import on a tile without neighbours (its own staging buffer: a 1 x 1 grid has no receive buffer), on a world attached to
    the library transport (refused), mi_world_step with several sub-steps on the caller's transport (refused: the exchange lies in between)."""
    sc = scenes.obb_pile(6, 3, 6, spacing=1.0)
    w = sc.populate(mi_lib.create_world(0))
    w.shard_enable(sharding._desc_for(sharding.tile_grid(sc, 1), 0))
    s = sc.settings()
    w.step_fixed(s, sc.dt, 1)
    empty = np.zeros(w.shard_message_bytes() // 4, np.float32)
    w.shard_import(empty)                                          # no neighbours, no records: accepted, nothing to apply
    st = w.shard_exchange_stats()
    assert st["num_neighbours"] == 0 and st["owned_bodies"] == sc.num_bodies and st["ghost_bodies"] == 0 and st["exchanges"] == 1
    several = capi.StepSettings(1, 120, 4, s.num_rigid_solver_iterations)
    with pytest.raises(capi.PhysicsError):
        w.step(several, 4.0 / 120.0)
    one = capi.StepSettings(1, 120, 1, s.num_rigid_solver_iterations)
    w.step(one, 1.0 / 120.0)
    if w.L.shard_library_transport_available():
        w.shard_attach_rccl(w.L.shard_unique_id())
        with pytest.raises(capi.PhysicsError):
            w.shard_import(empty)
        with pytest.raises(capi.PhysicsError):
            w.shard_set_axis_sums(np.zeros(9, np.uint64))
        w.step(several, 4.0 / 120.0)                               # the library transport exchanges inside every internal step
        w.shard_set_exact_seam(True, None)                         # exact seam through the library transport: one (here empty) send / receive group per sweep
        w.step(several, 4.0 / 120.0)
        assert w.seam_stats()["violations"] == 0
        with pytest.raises(capi.PhysicsError):
            w.shard_import_sweep(np.zeros(w.shard_sweep_message_bytes() // 4, np.float32))   # the library exchanges the sweeps itself
        w.shard_set_exact_seam(False, None)
        w.shard_detach_rccl()
    w.close()


def _rccl_rank(rank, world_size, port, out_dir, tiles_z, steps):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", rank))
    import d3d12renderer_amd as mi
    sc = scenes.obb_pile(12, 4, 8, spacing=1.0)
    desc = sharding.tile_grid(sc, world_size, tiles_z, 2.5)
    sw = sharding.ShardedWorld(sc.populate(mi.create_world(rank)), desc, rank, "rccl", dist)
    assert sw.transport == "rccl", sw.note
    s = sc.settings()
    owned = []
    for i in range(steps):
        sw.step(s, sc.dt)
        owned.append(sw.world.shard_counts()["owned_bodies"])
        if i == steps // 2:
            sw.rebalance()                                         # one load-balance round through the library (ncclAllReduce of the histograms)
    ents, st = sw.owned_states()
    stats = sw.world.shard_exchange_stats()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ents=ents, states=st, owned=np.asarray(owned), axis=sw.world.counts()["sorting_axis"],
             exchanges=stats["exchanges"], records=np.asarray(stats["records_sum"], np.uint64), borders=np.concatenate(sw.world.shard_get_borders(desc.tiles_x, desc.tiles_z)))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world_size,tiles_z", [(2, 1), (4, 2)], ids=["2 GPUs, x slabs", "4 GPUs, 2 x 2 tiles"])
def test_gpu_real_rccl_ranks_equal_virtual_ranks(mi_lib, tmp_path, world_size, tiles_z):
    """One process per GPU, the library's own transport: ncclSend / ncclRecv of the neighbour messages and the 72-byte ncclAllReduce of the
    sweep-axis statistics on every world's stream (csrc/world.hip shardExchange), a load-balance round through ncclAllReduce — compared, bit
    for bit, with the same ranks run as virtual ranks of this process on one GPU.  Needs as many visible devices as ranks (skipped otherwise:
    RCCL refuses two ranks on one device)."""
    import torch
    if torch.cuda.device_count() < world_size:
        pytest.skip(f"needs {world_size} visible GPUs, this box has {torch.cuda.device_count()}")
    import os
    import torch.multiprocessing as mp
    steps = 60
    port = 29700 + (os.getpid() % 2000) + world_size
    mp.spawn(_rccl_rank, args=(world_size, port, str(tmp_path), tiles_z, steps), nprocs=world_size, join=True)
    sc = scenes.obb_pile(12, 4, 8, spacing=1.0)
    ranks = _ranks(lambda: mi_lib.create_world(0), sc, world_size, tiles_z, 2.5)
    s = sc.settings()
    owned = [[] for _ in ranks]
    for i in range(steps):
        sharding.step_local(ranks, s, sc.dt)
        for r in ranks:
            owned[r.rank].append(r.world.shard_counts()["owned_bodies"])
        if i == steps // 2:
            sharding.rebalance_local(ranks)
    moved = 0
    for r in ranks:
        got = np.load(tmp_path / f"rank{r.rank}.npz")
        ents, st = r.owned_states()
        assert np.array_equal(got["owned"], np.asarray(owned[r.rank])), f"rank {r.rank}: owned bodies per step"
        assert np.array_equal(got["ents"], ents) and got["states"].tobytes() == st.tobytes(), f"rank {r.rank}: owned states"
        assert int(got["axis"]) == r.world.counts()["sorting_axis"]
        assert np.array_equal(got["borders"], np.concatenate(r.world.shard_get_borders(ranks[0].desc.tiles_x, ranks[0].desc.tiles_z)))
        assert int(got["exchanges"]) >= steps - 1
        moved += int(got["records"].sum())
    assert moved > 0, "no record ever crossed a link"


# ---------------------------------------------------------------------------------------------------------------- exact seam (include/mi_shard.h)
@pytest.mark.parametrize("make,num_ranks,z_slabs,margin", [(lambda: scenes.obb_pile(12, 4, 8, spacing=1.0), 3, False, 2.5), (lambda: scenes.ragdolls(4, 3), 2, False, 3.5),
                                                           (lambda: scenes.mixed_stack(10, 4, 10), 2, True, 2.5)],
                         ids=["pile, 3 x slabs", "ragdolls, 2 x slabs", "mixed stack, 2 z slabs"])
def test_gpu_exact_seam_ranks_equal_the_single_world_and_the_oracle(mi_lib, oracle_mod, make, num_ranks, z_slabs, margin):
    """Exact seam on the GPU: virtual ranks stepping side by side (one thread each, the per-sweep messages handed over inside the library's sweep
    callback) == ONE GPU world told the tiling == the ORACLE told the tiling, bit for bit, step after step.  The seam manifolds take the leading
    colours (k_emit_manifolds / k_color_round), every sweep is one launch followed by the hand-over (k_seam_sweep_pack / _unpack)."""
    sc = make()
    desc = sharding.tile_grid(sc, num_ranks, num_ranks if z_slabs else 1, margin)
    single = sc.populate(mi_lib.create_world(0)); single.set_seam_tiling(desc)
    ora = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL)); ora.set_seam_tiling(desc)
    ranks = [sharding.ShardedWorld(sc.populate(mi_lib.create_world(0)), desc, r, "local") for r in range(num_ranks)]
    s = sc.settings()
    ents = np.flatnonzero(sc.entities["kind"] != capi.ENTITY_STATIC).astype(np.uint32)
    seam_max = 0
    for i in range(60):
        single.step_fixed(s, sc.dt, 1); ora.step_fixed(s, sc.dt, 1)
        sharding.step_local_exact(ranks, s, sc.dt)
        ref = single.get_body_states(ents)
        assert single.counts() == ora.counts() and ref.tobytes() == ora.get_body_states(ents).tobytes(), f"step {i}: GPU world told the tiling vs the oracle told the tiling"
        assert sharding.gather_owned(ranks, len(ents)).tobytes() == ref.tobytes(), f"step {i}: exact-seam ranks vs the single world"
        assert single.seam_stats() == ora.seam_stats()
        seam_max = max(seam_max, single.seam_stats()["seam_manifolds"])
    assert seam_max > 0 and single.seam_stats()["violations"] == 0 and all(r.world.seam_stats()["violations"] == 0 for r in ranks)
    total, spec, _ = single.step_mode_stats()
    assert spec >= total - 3, "the single world told the tiling keeps its speculative fast path"
    w = sc.populate(mi_lib.create_world(0)); w.shard_enable(sharding._desc_for(sharding.tile_grid(sc, 4, 2, margin), 0))
    with pytest.raises(capi.PhysicsError):
        w.shard_set_exact_seam(True, None)                         # 2 x 2 tiles: slabs only
    w.close()


def _rccl_exact_rank(rank, world_size, port, out_dir, steps):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", rank))
    import d3d12renderer_amd as mi
    sc = scenes.obb_pile(12, 4, 8, spacing=1.0)
    desc = sharding.tile_grid(sc, world_size, 1, 2.5)
    sw = sharding.ShardedWorld(sc.populate(mi.create_world(rank)), desc, rank, "rccl", dist)
    assert sw.transport == "rccl", sw.note
    sw.enable_exact_seam()                                         # per-sweep ncclSend / ncclRecv on the world's stream
    s = sc.settings()
    for _ in range(steps):
        sw.step(s, sc.dt)
    ents, st = sw.owned_states()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ents=ents, states=st, violations=sw.world.seam_stats()["violations"])
    dist.barrier(); dist.destroy_process_group()


def test_gpu_exact_seam_real_rccl_ranks_equal_the_single_world(mi_lib, tmp_path):
    """One process per GPU, library transport: 20 small ncclSend / ncclRecv groups per step (one per sweep) — the union of what the ranks own equals
    the single GPU world told the tiling, bit for bit.  Needs 2 visible devices (skipped otherwise)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 visible GPUs, this box has {torch.cuda.device_count()}")
    import os
    import torch.multiprocessing as mp
    steps = 40
    mp.spawn(_rccl_exact_rank, args=(2, 29800 + (os.getpid() % 2000), str(tmp_path), steps), nprocs=2, join=True)
    sc = scenes.obb_pile(12, 4, 8, spacing=1.0)
    single = sc.populate(mi_lib.create_world(0)); single.set_seam_tiling(sharding.tile_grid(sc, 2, 1, 2.5))
    s = sc.settings()
    for _ in range(steps):
        single.step_fixed(s, sc.dt, 1)
    seen = {}
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert int(got["violations"]) == 0
        for e, st in zip(got["ents"], got["states"]):
            assert int(e) not in seen; seen[int(e)] = st
    ents = np.asarray(sorted(seen), np.uint32)
    assert len(ents) == sc.num_bodies and np.stack([seen[int(e)] for e in ents]).tobytes() == single.get_body_states(ents).tobytes()
