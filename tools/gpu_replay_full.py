"""Replay of the reference's constraint order at the LARGEST sizes the reference's 16-bit collider indices allow (cfg2 / cfg3: 16 384 bodies, cfg4: 256 ragdolls,
cfg5: 64 vehicles), >= 200 steps, free-running, every step compared bit for bit with oracle/_ref/libref.so (tests/helpers.py: replay_reference_order) — the sizes
the regular suite leaves out because the replay solves sequentially (k_contact_solve_serial / k_joint_solve_serial: one lane).  One line per scene."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
import oracle
from helpers import replay_reference_order

SCENES = {
    "cfg3 16384 boxes": (lambda: scenes.obb_pile(32, 16, 32), 240),
    "cfg2 16384 mixed": (lambda: scenes.mixed_stack(32, 16, 32), 200),
    "cfg4 256 ragdolls": (lambda: scenes.ragdolls(16, 16), 200),
    "cfg5 64 vehicles": (lambda: scenes.vehicles(8, 8), 200),
}
which = sys.argv[1:] or list(SCENES)
for name in which:
    make, steps = SCENES[name]
    t = time.time()
    try:
        most = replay_reference_order(lambda: mi.create_world(0), lambda: oracle.create_reference_world(), make(), steps)
        print(f"[replay] {name}: {steps} steps, every count, contact, pose and velocity bit equal to the reference's; most contacts in a step {most}; {time.time() - t:.0f} s", flush=True)
    except AssertionError as e:
        print(f"[replay] {name}: MISMATCH {str(e)[:300]} after {time.time() - t:.0f} s", flush=True)
