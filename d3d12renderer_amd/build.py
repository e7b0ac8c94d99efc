"""Builds the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libmi_physics.so"
SOURCES = [CSRC / "world.hip"]
HEADERS = [CSRC / n for n in ("dmath.hpp", "narrow.hpp", "kernels.hpp", "kernels_common.hpp", "kernels_broad.hpp", "kernels_narrow.hpp", "kernels_integrate.hpp", "kernels_schedule.hpp", "kernels_contacts.hpp", "kernels_solve.hpp", "kernels_shard.hpp", "kernels_scan.hpp", "gjk.hpp", "joints.hpp", "heightmap.hpp", "cloth.hpp", "launcher.hpp", "knobs.hpp", "world_setup.inc", "world_step.inc", "world_joints.inc", "world_capi.inc", "world_shard.inc", "world_state.inc")] + \
          [HERE.parent / "include" / n for n in ("mi_physics.h", "mi_constraints.h", "mi_shard.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-fvisibility=hidden",
         "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
         "-Wno-unused-result", "-Wno-unused-function",
         "-Wno-inline-asm"]   # (the persistent solver names v255 in a clobber list on purpose: its kernels are capped at 184 allocatable VGPRs and keep rows in the rest)


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS)


LEARNING_LIB = HERE / "libPhysics-Lib.so"      # the reference's learning DLL ("Physics-Lib.dll") over libmi_physics.so
LEARNING_SRC = CSRC / "learning.cpp"
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math", "-fopenmp"]


def build_learning(force=False, verbose=False):
    """libPhysics-Lib.so: host-only C++ (g++) linked against libmi_physics.so next to it ($ORIGIN rpath)."""
    deps = [LEARNING_SRC, LIB] + HEADERS[-2:]
    if not force and LEARNING_LIB.exists() and all(p.stat().st_mtime <= LEARNING_LIB.stat().st_mtime for p in deps):
        return LEARNING_LIB
    cmd = [os.environ.get("CXX", "g++"), *HOST_FLAGS, str(LEARNING_SRC), "-o", str(LEARNING_LIB), "-L", str(HERE), "-l:libmi_physics.so", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("g++ failed (learning library)")
    return LEARNING_LIB


def build(force=False, verbose=False):
    lib = _build_physics(force, verbose)
    build_learning(force, verbose)
    return lib


def _build_physics(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), *FLAGS, "-I", str(HERE.parent / "include"), *map(str, SOURCES), "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed")
    return LIB


def device_asm(out_path):
    """gfx950 assembly of the device code (hipcc -S --cuda-device-only): inline-asm blocks stay delimited by #ASMSTART / #ASMEND,
    which tests/test_capi_symbols.py uses to check what the compiler did around the persistent solver's hand-placed registers."""
    flags = [f for f in FLAGS if f not in ("-shared", "-fPIC", "-fvisibility=hidden")]
    cmd = [hipcc(), *flags, "-I", str(HERE.parent / "include"), "-S", "--cuda-device-only", *map(str, SOURCES), "-o", str(out_path)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc -S failed")
    return out_path


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
