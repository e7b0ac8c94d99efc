"""A slice of the differential fuzzer (tools/gpu_fuzz.py) in the suite: random worlds — every collider type, compound bodies, deep overlaps, joints, terrain, triggers and
force fields, events, uneven frame times; state writes, deletions, spawns, in-place checkpoints between the steps — stepped in the oracle and on the GPU under the default
environment and one of the library's other paths: every count, event and state bit equal.  (The long runs: profiles/r06_fuzz_*.json.  The reference has no such test.)"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import gpu_fuzz   # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("first, count, scale, steps", [(0, 120, 1, 40), (7000, 120, 1, 40), (100, 6, 20, 25)], ids=["seeds-0", "seeds-7000", "large-worlds"])
def test_gpu_random_worlds_equal_the_oracle(mi_lib, oracle_mod, first, count, scale, steps):
    bad = []
    for seed in range(first, first + count):
        r = gpu_fuzz.run_seed(seed, steps, oracle_mod, scale=scale)
        if r:
            bad.append(r)
    assert not bad, bad[:3]


@pytest.mark.gpu
def test_gpu_random_worlds_cut_into_tiles_equal_the_oracles_ranks(mi_lib, oracle_mod):
    """The sharded sibling (tools/gpu_fuzz_sharded.py): the same random worlds as 2-4 virtual ranks on the GPU and in the oracle's sharding mirror — local counts, owned counts,
    owned states bit for bit (localized force fields acting on ghosts included: what the long run found in the mirror)."""
    import gpu_fuzz_sharded
    bad = [r for r in (gpu_fuzz_sharded.run_seed(seed, 30, oracle_mod, 4) for seed in range(30, 70)) if r]
    assert not bad, bad[:3]


def test_fuzz_world_generator_is_deterministic_and_valid(oracle_mod):
    """(CPU) the same seed gives the same world and plan; what it generates is a world the oracle steps without a non-finite state in the first steps."""
    import numpy as np
    for seed in (1, 153, 5039):
        a, ba, _ = gpu_fuzz.make_world_description(seed); b, bb, _ = gpu_fuzz.make_world_description(seed)
        assert a.entities.tobytes() == b.entities.tobytes() and a.colliders.tobytes() == b.colliders.tobytes() and (ba == bb).all()
        pa, pb = gpu_fuzz.plan_actions(seed, 10), gpu_fuzz.plan_actions(seed, 10)
        assert all(x[0] == y[0] and x[1] == y[1] and (x[2] == y[2]).all() and (x[3] == y[3]).all() for x, y in zip(pa, pb))
        q = a.colliders["shape"][a.colliders["type"] >= 4, :4]                      # OBB / hull rotations are unit quaternions
        assert np.allclose((q * q).sum(axis=1), 1.0, atol=1e-5)
        w = a.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
        w.step_fixed(a.settings(), a.dt, 2)
        assert np.isfinite(w.get_body_states(ba)).all()
        w.close()
