"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
It builds oracle/_build/liboracle.so from the C++ restatement (oracle/Makefile) and wraps it with
the same ctypes binding class the product uses (the oracle exports the ABI with an `ora_` prefix).
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

from d3d12renderer_amd import capi

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "liboracle.so"

ORDER_REFERENCE = 0   # SAP sweep order + sequential PGS: the reference's scalar path as written
ORDER_CANONICAL = 1   # same arithmetic, pairs/colours ordered like the GPU schedule (replayed sequentially)


def build(force=False):
    srcs = [HERE / f for f in ("ora_world.cpp", "ora_det.cpp", "ora_narrow.cpp", "ora_gjk.cpp", "ora_joints.cpp", "ora_heightmap.cpp", "ora_cloth.cpp", "ora_math.h", "ora_world.h")]
    srcs += [HERE.parent / "include" / "mi_physics.h", HERE.parent / "include" / "mi_constraints.h"]
    if force or not LIB.exists() or any(s.stat().st_mtime > LIB.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
    return LIB


LEARNING_LIB = HERE / "_build" / "liboracle_learning.so"


def build_learning():
    """The product's learning environment code (d3d12renderer_amd/csrc/learning.cpp) compiled against the ORACLE's C ABI, so the
    parity tests can run identical environment code over both physics backends.  Test infrastructure only."""
    build()
    src = HERE.parent / "d3d12renderer_amd" / "csrc" / "learning.cpp"
    if not LEARNING_LIB.exists() or any(p.stat().st_mtime > LEARNING_LIB.stat().st_mtime for p in (src, LIB, HERE / "ora_learning_backend.h")):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                        f'-DMI_LEARNING_BACKEND_HEADER="{HERE / "ora_learning_backend.h"}"', str(src), "-o", str(LEARNING_LIB), "-L", str(LIB.parent), "-l:liboracle.so", "-Wl,-rpath,$ORIGIN"],
                       check=True, capture_output=True)
    return LEARNING_LIB


_library = None


def library():
    global _library
    if _library is None:
        build()
        _library = capi.Library(LIB, prefix="ora_")
    return _library


def create_world(order=ORDER_REFERENCE):
    L = library()
    h = C.c_void_p()
    L.check(L.fn("world_create")(C.c_int(order), C.byref(h)), "world_create")
    return capi.World(L, h)


# ---------------------------------------------------------------------------------------------------------------------
# oracle/_ref: the REFERENCE's own sources compiled here (oracle/refbuild/build_ref.py) — what the restatement is pinned to.
REF_LIB = HERE / "_ref" / "libref.so"
_ref_library = None


def reference_available():
    """True when libref.so exists or can be built (it needs /root/reference; the prebuilt library travels to the GPU box)."""
    from oracle.refbuild import build_ref
    return REF_LIB.exists() or (build_ref.REFERENCE_ROOT / "src").exists()


def build_reference(force=False):
    """Both builds of the reference: libref.so (strict, the pin) and libref_fast.so (the reference's own AVX2 + fast-math flags, timing only)."""
    from oracle.refbuild import build_ref
    build_ref.build(force=force, variant="fast")
    return build_ref.build(force=force)


def build_binding():
    """oracle/_ref/binding_check (+ _oracle): the reference-side backend stub (oracle/refbuild/binding/physics_mi355x.cpp) compiled against the reference's
    real headers together with the reference's scene / physics code, linked to libmi_physics.so (GPU) or to this oracle (CPU).  Returns the two paths
    (prebuilt ones when /root/reference is not mounted)."""
    from oracle.refbuild.binding import build_binding as B
    build()
    return B.build(backend="product"), B.build(backend="oracle")


def reference_library():
    global _ref_library
    if _ref_library is None:
        build_reference()
        _ref_library = capi.Library(REF_LIB, prefix="ref_")
    return _ref_library


_ref_fast_library = None


def create_reference_world(simd=False, fast=False):
    """A world stepped by the reference's physicsStep itself (scalar path, or its AVX2 path with simd=True).
    fast=True uses libref_fast.so (timing build, not bit-comparable)."""
    global _ref_fast_library
    if fast:
        if _ref_fast_library is None:
            build_reference()
            _ref_fast_library = capi.Library(REF_LIB.with_name("libref_fast.so"), prefix="ref_")
        L = _ref_fast_library
    else:
        L = reference_library()
    h = C.c_void_p()
    L.check(L.fn("world_create")(C.c_int(1 if simd else 0), C.byref(h)), "world_create")
    return capi.World(L, h)
