"""Sharded worlds on the GPU (include/mi_shard.h, csrc k_shard_*): several ranks of one scene as several worlds of ONE process on one
GPU, the neighbour messages moved through the caller's-transport entry points (mi_world_shard_export / _import).  The CPU oracle
mirrors the sharding, so every rank is compared with its oracle twin bit for bit; tests/test_distributed.py shows that real
processes over a real transport give exactly what such virtual ranks give."""
import numpy as np
import pytest

from d3d12renderer_amd import scenes, sharding

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
    w.shard_detach_rccl()                                             # back to the caller's transport (what a rank does when a peer could not attach)
    w.step_fixed(s, sc.dt, 1); plain.step_fixed(s, sc.dt, 1)
    assert w.physics_transforms()[0].tobytes() == plain.physics_transforms()[0].tobytes()
    w.close()
