"""MI355X-native rigid-body stepper: drop-in for the `src/physics` step path of pkurth/D3D12Renderer.

The product is the HIP shared library `libmi_physics.so` (C ABI in include/mi_physics.h); this
package is its Python host-side mirror (ctypes).  There is no CPU fallback: if the library is
missing or no HIP device is visible, creating a world raises.
"""
import ctypes as C
from pathlib import Path

from . import capi, scenes  # noqa: F401
from .capi import StepSettings, PhysicsError  # noqa: F401

import os

# MI_PHYSICS_LIB: development override (A/B-testing another build of the same HIP library); never a CPU path
LIB_PATH = Path(os.environ["MI_PHYSICS_LIB"]).resolve() if os.environ.get("MI_PHYSICS_LIB") else Path(__file__).resolve().parent / "libmi_physics.so"
_library = None


def library():
    """Loads the in-tree HIP extension (built by `python -m d3d12renderer_amd.build` / __graft_entry__.build())."""
    global _library
    if _library is None:
        if not LIB_PATH.exists():
            raise PhysicsError(f"{LIB_PATH} is missing: build it with `python -m d3d12renderer_amd.build` "
                               "(the stepper has no CPU fallback)")
        _library = capi.Library(LIB_PATH, prefix="mi_")
    return _library


def create_world(device=0):
    """physics_world constructor <-> game_scene + memory_arena (src/physics/physics.cpp:1205)."""
    L = library()
    desc = capi.WorldDesc(device, 0)
    h = C.c_void_p()
    L.check(L.fn("world_create")(C.byref(desc), C.byref(h)), "world_create")
    return capi.World(L, h)
