"""Generates tests/golden/*.npz from the CPU oracle (canonical order).

The reference ships no golden vectors and cannot be built here, so these are
ORACLE-generated regression vectors: they pin (a) the oracle against silent drift and (b) the HIP
path bit-for-bit on the GPU box, where the oracle is rebuilt from source and re-checked against
them first.  Run: python tools/make_golden.py
"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle
from d3d12renderer_amd import scenes

CASES = {
    "cfg1_spheres_216": (lambda: scenes.sphere_drop(6), 140),
    "cfg2_mixed_96": (lambda: scenes.mixed_stack(4, 6, 4), 90),
    "cfg3_obb_160": (lambda: scenes.obb_pile(4, 10, 4, spacing=1.0), 90),
    "zoo_all_shapes_144": (lambda: scenes.shape_zoo(), 160),
    "cfg4_ragdolls_4": (lambda: scenes.ragdolls(2, 2), 150),
    "joint_zoo_112": (lambda: scenes.joint_zoo(), 150),
    "cfg5_vehicles_2": (lambda: scenes.vehicles(2, 1), 150),
}


def run_case(make, steps, order):
    sc = make()
    w = sc.populate(oracle.create_world(order))
    s = sc.settings()
    counts = []
    for _ in range(steps):
        w.step_fixed(s, sc.dt, 1)
        c = w.counts()
        counts.append([c["num_broadphase_overlaps"], c["num_collisions"], c["num_contacts"], c["num_colors"], c["sorting_axis"]])
    p, q = w.physics_transforms()
    v, a = w.velocities()
    return dict(pos=p, rot=q, lin=v, ang=a, counts=np.asarray(counts, np.uint32))


if __name__ == "__main__":
    out = ROOT / "tests" / "golden"
    out.mkdir(parents=True, exist_ok=True)
    for name, (make, steps) in CASES.items():
        for order, tag in ((oracle.ORDER_CANONICAL, "canonical"), (oracle.ORDER_REFERENCE, "reference")):
            r = run_case(make, steps, order)
            np.savez_compressed(out / f"{name}_{tag}.npz", steps=np.uint32(steps), **r)
            print(name, tag, r["counts"][-1])
