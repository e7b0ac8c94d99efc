"""Run as a script (tests/test_learning.py spawns it: the environment library keeps one world per process): the product's
learning environment code (csrc/learning.cpp, compiled over the oracle's ABI in the REFERENCE's constraint order) against the
reference's own DLL entry points — getPhysicsStateSize / getPhysicsActionSize / getPhysicsRanges / resetPhysics / updatePhysics
of src/learning/learned_locomotion.cpp:395-489, compiled into oracle/_ref/libref.so together with humanoid_ragdoll
(src/physics/ragdoll.cpp).  Two test hooks are patched into the reference copy (oracle/refbuild/build_ref.py): the push RNG's
state can be set (the original seeds it with time(0)) and the step can be told to take the scalar path (the original steps with
the default physics_settings, i.e. its AVX2 path)."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

os.environ["MI_LEARNING_ORACLE_ORDER"] = "0"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import oracle                                        # noqa: E402
from d3d12renderer_amd.learning import PhysicsDLL    # noqa: E402

mine = PhysicsDLL(oracle.build_learning())
oracle.build_reference()
ref = C.CDLL(str(oracle.REF_LIB))
F = C.POINTER(C.c_float)
ref.updatePhysics.argtypes = (F, F, F); ref.resetPhysics.argtypes = (F,); ref.getPhysicsRanges.argtypes = (F, F, F, F)
ref.refLearningConfigure.argtypes = (C.c_ulonglong, C.c_int)
fp = lambda a: a.ctypes.data_as(F)                   # noqa: E731
assert (ref.getPhysicsStateSize(), ref.getPhysicsActionSize()) == (mine.state_size, mine.action_size) == (66, 27)
r = [np.zeros(66, np.float32), np.zeros(66, np.float32), np.zeros(27, np.float32), np.zeros(27, np.float32)]
ref.getPhysicsRanges(*map(fp, r))
m = mine.ranges()
assert all(a.tobytes() == b.tobytes() for a, b in zip(r, m)), "getPhysicsRanges"
amin, amax = m[2], m[3]
steps = pushes_seen = 0
for seed, scale in ((12345, 0.15), (7, 0.05), (99, 0.6)):
    mine.seed(seed); ref.refLearningConfigure(C.c_ulonglong((seed + 0x632BE59BD9B4E019) % 2 ** 64), 0)      # environment 0's RNG stream
    mine.reset(); ref.resetPhysics(fp(np.zeros(66, np.float32)))
    rng = np.random.default_rng(seed)
    for i in range(250):
        a = (rng.uniform(-1, 1, 27) * scale * (amax - amin)).astype(np.float32)
        sm, rm, dm = mine.step(a)
        sr = np.zeros(66, np.float32); rr = np.zeros(1, np.float32)
        dr = ref.updatePhysics(fp(a), fp(sr), fp(rr))
        assert sm.tobytes() == sr.tobytes(), f"seed {seed} step {i}: state"
        assert np.float32(rm).tobytes() == rr.tobytes() and int(dm) == int(dr), f"seed {seed} step {i}: reward / done"
        steps += 1
        if dm:
            break
print(f"REFERENCE_DLL_OK {steps} steps")
