"""The checks that used to run only as end-of-round soaks (tools/gpu_sync_vs_spec.py, tools/gpu_replay_full.py), as part of the GPU suite: both illegal memory accesses of
rounds 3 and 4 lived in paths only these reach — the synchronous step at full size (a world's first step, and every re-run of a speculative step), and the reference-order
replay at the largest size the reference's 16-bit collider indices allow.
"""
import hashlib

import pytest

from d3d12renderer_amd import scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,make,steps", [
    ("cfg1 4096 spheres", lambda: scenes.sphere_drop(16), 200),
    ("cfg2 65536 mixed", lambda: scenes.mixed_stack(64, 16, 64), 100),
    ("cfg3 262144 boxes", lambda: scenes.obb_pile(128, 16, 128), 100),
    ("cfg4 1024 ragdolls", lambda: scenes.ragdolls(32, 32), 150),
    ("cfg5 256 vehicles", lambda: scenes.vehicles(16, 16), 150),
    ("terrain 65536", lambda: scenes.terrain_big(), 120),
    ("zones 6912 (triggers, force fields)", lambda: scenes.zones(48, 3, 48), 150),
], ids=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "terrain", "zones"])
def test_gpu_synchronous_steps_equal_speculative_steps_at_full_size(mi_lib, monkeypatch, name, make, steps):
    """Every BASELINE configuration at full size, stepped speculatively (the default: launches sized from the previous step, one read-back per step) and synchronously
    (MI_ASYNC=0: exact sizes read back inside every step — the path of a world's first step and of every re-run): the two must end in the same bits."""
    res = {}
    for mode in ("speculative", "synchronous"):
        if mode == "synchronous":
            monkeypatch.setenv("MI_ASYNC", "0")
        else:
            monkeypatch.delenv("MI_ASYNC", raising=False)
        sc = make(); w = sc.populate(mi_lib.create_world(0))
        w.step_fixed(sc.settings(), sc.dt, steps)
        p, q = w.physics_transforms(); v, a = w.velocities()
        res[mode] = (hashlib.sha1(p.tobytes() + q.tobytes() + v.tobytes() + a.tobytes()).hexdigest(), w.counts())
        modes = w.step_mode_stats()
        w.close()
        if mode == "synchronous":
            assert modes[1] == 0, f"{name}: MI_ASYNC=0 must not step speculatively {modes}"
    assert res["speculative"] == res["synchronous"], name


def test_gpu_reference_order_replay_at_the_reference_index_limit(mi_lib, oracle_mod):
    """cfg2 at 16 384 bodies (the reference's u16 collider indices end at 65 535), told the reference's own constraint order step by step: every count, contact, pose and
    velocity bit equals oracle/_ref/libref.so's (the reference compiled here).  40 steps from the lattice; tools/gpu_replay_full.py runs 200+ steps of all four scenes."""
    if not oracle_mod.REF_LIB.exists() and not oracle_mod.reference_available():
        pytest.skip("oracle/_ref/libref.so is not here and /root/reference is not mounted")
    from helpers import replay_reference_order
    most = replay_reference_order(lambda: mi_lib.create_world(0), lambda: oracle_mod.create_reference_world(), scenes.mixed_stack(32, 16, 32), 40)
    assert most > 10000, most
