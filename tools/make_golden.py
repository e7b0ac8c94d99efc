"""Generates tests/golden/*.npz.

  *_reference.npz   outputs of THE REFERENCE ITSELF: the reference's own physics sources compiled into oracle/_ref/libref.so
                    (oracle/refbuild/build_ref.py) and stepped through its own physicsStep.  The reference ships no golden vectors of
                    its own; these are the vectors the oracle (reference order) has to reproduce bit for bit on any machine,
                    including the GPU box, where /root/reference does not exist.
  *_canonical.npz   the oracle replaying the GPU's canonical schedule (same arithmetic, colour-major PGS order): what the HIP path
                    has to reproduce bit for bit.
Run here (needs /root/reference): python tools/make_golden.py
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


def run_case(make, steps, order, from_reference=False):
    sc = make()
    w = sc.populate(oracle.create_reference_world() if from_reference else oracle.create_world(order))
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
            from_ref = tag == "reference"
            r = run_case(make, steps, order, from_reference=from_ref)
            source = "reference: oracle/_ref/libref.so = /root/reference sources, own physicsStep" if from_ref else "oracle, canonical schedule"
            np.savez_compressed(out / f"{name}_{tag}.npz", steps=np.uint32(steps), source=np.array(source), **r)
            print(name, tag, r["counts"][-1])
