"""BASELINE.json's five configurations AT FULL SIZE, GPU (through the C ABI) against the oracle replaying the canonical schedule.

cfg1 / cfg4 / cfg5 are small enough for the oracle to run from the first step.  For cfg2 (65 536 bodies) and cfg3 (262 144
bodies, the headline workload) the GPU settles the scene, writes a checkpoint (mi_world_save_checkpoint: body states, colour
history, SAP axis, accumulators), the oracle loads that blob (ora_world_load_checkpoint) and both continue: every count and every
pose / velocity bit of every step must agree.  The oracle itself is pinned to the reference's own code (tests/test_reference_pin.py).
"""
import numpy as np
import pytest

from d3d12renderer_amd import scenes

pytestmark = pytest.mark.gpu


def _same(g, o, tag):
    assert g.counts() == o.counts(), tag
    for a, b in zip(g.physics_transforms() + g.velocities(), o.physics_transforms() + o.velocities()):
        assert a.tobytes() == b.tobytes(), tag


@pytest.mark.parametrize("name,make,steps", [
    ("cfg1 4096 spheres", lambda: scenes.sphere_drop(16), 240),
    ("cfg4 1024 ragdolls", lambda: scenes.ragdolls(32, 32), 150),
    ("cfg5 256 vehicles", lambda: scenes.vehicles(16, 16), 150),
    ("6912 bodies, triggers + force fields", lambda: scenes.zones(48, 3, 48), 200),      # interactions ordered and applied on the device, speculative steps
], ids=["cfg1", "cfg4", "cfg5", "zones"])
def test_gpu_full_size_from_the_first_step(mi_lib, oracle_mod, name, make, steps):
    sc = make()
    g = sc.populate(mi_lib.create_world(0)); o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    if "triggers" in name:
        g.enable_events(True); o.enable_events(True)
    s = sc.settings()
    for i in range(0, steps, 10):
        g.step_fixed(s, sc.dt, 10); o.step_fixed(s, sc.dt, 10)
        assert g.counts() == o.counts(), f"{name}: step {i + 10}"
        if "triggers" in name:
            eg, eo = g.poll_events(), o.poll_events()
            assert eg.tobytes() == eo.tobytes(), f"{name}: events of steps {i + 1}..{i + 10}"
    _same(g, o, name)
    assert g.counts()["num_contacts"] > 0
    if "triggers" in name:
        total, speculative, _ = g.step_mode_stats()
        assert speculative >= total - 3, "trigger / force-field scenes step speculatively (interactions stay on the device)"


@pytest.mark.parametrize("name,make,settle,cont", [
    ("cfg2 65536 mixed", lambda: scenes.mixed_stack(64, 16, 64), 240, 12),
    ("cfg3 262144 boxes", lambda: scenes.obb_pile(128, 16, 128), 240, 10),
    ("65536 bodies on heightmap terrain", lambda: scenes.terrain_big(), 260, 8),
], ids=["cfg2", "cfg3", "terrain"])
def test_gpu_full_size_continues_from_a_checkpoint(mi_lib, oracle_mod, name, make, settle, cont):
    sc = make()
    g = sc.populate(mi_lib.create_world(0))
    s = sc.settings()
    g.step_fixed(s, sc.dt, settle)
    blob = g.save_checkpoint()
    o = sc.populate(oracle_mod.create_world(oracle_mod.ORDER_CANONICAL))
    o.load_checkpoint(blob)
    for a, b in zip(g.physics_transforms() + g.velocities(), o.physics_transforms() + o.velocities()):
        assert a.tobytes() == b.tobytes(), f"{name}: state after loading the checkpoint"
    nb = sc.num_bodies
    assert g.counts()["num_contacts"] > (2 * nb if "terrain" not in name else nb), "not a settled scene"
    for i in range(cont):
        g.step_fixed(s, sc.dt, 1); o.step_fixed(s, sc.dt, 1)
        _same(g, o, f"{name}: step {settle + i + 1}")
    print(name, g.counts())
