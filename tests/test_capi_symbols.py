"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/*.h declares.
No compute is called here; without a device, world creation must fail loudly (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path
import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    names = []
    for h in (ROOT / "include").glob("*.h"):
        names += re.findall(r"MI_API\s+[\w\s\*]+?\b(mi_\w+)\s*\(", h.read_text())
    return sorted(set(names))


def test_header_declares_the_step_api():
    syms = declared_symbols()
    for required in ("mi_world_create", "mi_world_destroy", "mi_entity_create", "mi_collider_add", "mi_constraint_create",
                     "mi_world_step", "mi_world_step_fixed", "mi_world_get_transforms", "mi_world_get_counts", "mi_world_get_contacts"):
        assert required in syms


def test_library_exports_every_declared_symbol(mi_lib):
    L = mi_lib.library()
    missing = [s for s in declared_symbols() if not hasattr(L.lib, s)]
    assert not missing, missing


def test_learning_library_exports_the_reference_abi(mi_lib):
    """libPhysics-Lib.so exports every function include/mi_learning.h declares (the five of learned_locomotion.cpp:395-489 first)."""
    import ctypes
    from d3d12renderer_amd import learning
    names = re.findall(r"MI_LEARNING_API\s+[\w\s\*]+?\b(\w+)\s*\(", (ROOT / "include" / "mi_learning.h").read_text())
    assert names[:5] == ["getPhysicsStateSize", "getPhysicsActionSize", "getPhysicsRanges", "resetPhysics", "updatePhysics"]
    lib = ctypes.CDLL(str(learning.LIB_PATH))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.getPhysicsStateSize() == 66 and lib.getPhysicsActionSize() == 27


def test_no_cpu_fallback_without_device(mi_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(mi_lib.PhysicsError) as e:
        mi_lib.create_world()
    assert "no HIP device" in str(e.value) or "status -2" in str(e.value)


def test_struct_sizes_match_header():
    from d3d12renderer_amd import capi
    assert capi.entity_desc.itemsize == 17 * 4          # mi_entity_desc
    assert capi.collider_desc.itemsize == 18 * 4        # mi_collider_desc
    assert capi.contact_dtype.itemsize == 12 * 4        # mi_contact
    assert C.sizeof(capi.StepSettings) == 16 and C.sizeof(capi.StepCounts) == 32 and C.sizeof(capi.StageTimes) == 36
    assert capi.hinge_constraint.itemsize == 104 and capi.cone_twist_constraint.itemsize == 120   # = reference sizes
    assert capi.distance_constraint.itemsize == 28 and capi.ball_constraint.itemsize == 24
    assert capi.fixed_constraint.itemsize == 40 and capi.slider_constraint.itemsize == 72
    assert C.sizeof(capi.ShardDesc) == 40 and capi.SHARD_RECORD_FLOATS == 14                       # mi_shard_desc, MI_SHARD_RECORD_FLOATS


def _build_facade(tmp_path, mi_lib):
    import subprocess
    exe = tmp_path / "facade_smoke"
    libdir = ROOT / "d3d12renderer_amd"
    subprocess.run(["g++", "-std=c++17", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "facade_smoke.cpp"), "-o", str(exe),
                    f"-L{libdir}", "-lmi_physics", f"-Wl,-rpath,{libdir}"], check=True)
    return exe


def test_cpp_facade_compiles_and_links(tmp_path, mi_lib):
    """include/physics_world.hpp (physics_world::step/addRigidBody/addConstraint, physicsStep, asXxx) builds with a plain
    C++17 compiler against the C ABI; without a GPU it must fail loudly, not fall back."""
    import subprocess
    import torch
    exe = _build_facade(tmp_path, mi_lib)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0 and "facade ok" in r.stdout
    else:
        assert r.returncode == 1 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_cpp_facade_runs_on_gpu(tmp_path, mi_lib):
    import subprocess
    exe = _build_facade(tmp_path, mi_lib)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "facade ok" in r.stdout, r.stdout + r.stderr


def test_persistent_solver_keeps_its_acc_registers_to_itself(tmp_path, mi_lib):
    """k_contact_solve_persist prefetches the next tile's constraint rows with inline-asm loads into the FIXED accumulator
    registers a160..a255, keeps the rows of a wave's first tiles RESIDENT in a0..a143 (loaded once; handed to the ring by
    v_accvgpr_mov_b32) and reads the ring back with inline asm after an explicit wait.  That is only sound if the compiler never
    touches an accumulator register itself: every reference to one must sit inside an #ASMSTART/#ASMEND block (no spill copies),
    nothing may go to scratch, and loads / moves / read-backs come in whole contacts (6 loads, 24 moves, 24 reads)."""
    from d3d12renderer_amd import build
    asm = build.device_asm(tmp_path / "device.s").read_text().split("\n")
    starts = [i for i, l in enumerate(asm) if re.match(r"^_ZN2mi23k_contact_solve_persistILb[01]ELb[01]ELb[01]EE.*:", l)]
    assert len(starts) == 6, "variants: slot data / impulses in LDS or not, XCD-partitioned or not"
    for st in starts:
        capped = bool(re.match(r"^_ZN2mi23k_contact_solve_persistILb1E", asm[st]))   # slot data in LDS: 208 allocatable VGPRs, two more resident positions in v208..v255
        in_asm, stray, ring_loads, res_loads, vres_loads, reads, moves, vmoves = False, [], 0, 0, 0, 0, 0, 0
        i = st
        num = lambda x: int(x, 0)    # ("n" operands print as hex from 160 on)
        while ".amdhsa_kernel" not in asm[i]:
            line = asm[i]; i += 1
            if "#ASMSTART" in line:
                in_asm = True; continue
            if "#ASMEND" in line:
                in_asm = False; continue
            code = line.split(";")[0]
            if in_asm:
                for lo in re.findall(r"global_load_dwordx4 a\[(0x[0-9a-f]+|\d+):", code):
                    if num(lo) >= 160: ring_loads += 1
                    else: res_loads += 1
                for lo in re.findall(r"global_load_dwordx4 v\[(0x[0-9a-f]+|\d+):", code):
                    if num(lo) >= 208:          # (below: the body / poll loads, into registers the compiler chose)
                        assert capped, lo
                        vres_loads += 1
                reads += len(re.findall(r"v_accvgpr_read_b32", code))
                for dst, src in re.findall(r"v_accvgpr_mov_b32 a\[?(0x[0-9a-f]+|\d+)\]?, a\[?(0x[0-9a-f]+|\d+)\]?", code):
                    assert num(dst) >= 160 and num(src) < 144 and (num(dst) - 160) % 24 == num(src) % 24, (dst, src)
                    moves += 1
                for dst, src in re.findall(r"v_accvgpr_write_b32 a\[?(0x[0-9a-f]+|\d+)\]?, v\[?(0x[0-9a-f]+|\d+)\]?", code):
                    assert capped and num(dst) >= 160 and num(src) >= 208 and (num(dst) - 160) % 24 == (num(src) - 208) % 24, (dst, src)
                    vmoves += 1
            elif "scratch_" in code or "v_accvgpr" in code or re.search(r"\ba\[?\d+", code):
                stray.append(line.strip())
            elif capped and any(int(x) >= 208 for x in re.findall(r"\bv\[?(\d+)", code) + re.findall(r"\bv\[\d+:(\d+)\]", code)):
                stray.append(line.strip())
        assert not stray, stray[:5]
        # the prefetch is inlined once per call site (24 loads each); the read-back once per contact count (24 moves per contact); six resident positions of six loads
        assert ring_loads > 0 and ring_loads % 24 == 0 and reads > 0 and reads % 24 == 0 and res_loads == 36 and moves > 0 and moves % 24 == 0, (ring_loads, res_loads, reads, moves)
        assert (vres_loads, vmoves > 0, vmoves % 24) == ((12, True, 0) if capped else (0, False, 0)), (vres_loads, vmoves)
        if re.match(r"^_ZN2mi23k_contact_solve_persistILb0ELb[01]ELb1E", asm[st]):
            # slot data not in LDS (impulses in LDS): a RESIDENT next tile leaves the three loads of its slot data in flight across the first tag check (`s_waitcnt vmcnt(3)`), which is only
            # sound while the compiler issues exactly those three loads between the tile's four body loads and the counted waits — checked here for every contact count
            seen = 0; j = st
            while ".amdhsa_kernel" not in asm[j]:
                if re.search(r"global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off sc1", asm[j]) and "sc1" in asm[j + 3] and "#ASMEND" in asm[j + 4]:   # the four body loads of a visit
                    k, inside, mine = j + 5, False, 0
                    while not (inside and "s_waitcnt vmcnt(" in asm[k] and "vmcnt(4)" not in asm[k]):
                        if "#ASMSTART" in asm[k]: inside = True
                        elif "#ASMEND" in asm[k]: inside = False
                        elif not inside and asm[k].strip().startswith("global_load"): mine += 1
                        k += 1
                    assert mine == 3, (asm[st][:60], j, mine)
                    seen += 1; j = k
                j += 1
            assert seen == 4, seen




def test_tile_to_xcd_assignment_is_a_partition(mi_lib):
    """The XCD-partitioned solver gives tile tl of a bin's nt tiles to XCD floor(8 tl / nt) (short bins round-robin) and finds a
    tile's place in that XCD's list from closed forms.  For every bin size up to 70 tiles (and a few large ones) and several bin
    indices: the eight shares add up to nt, (owner, rank) enumerates each share exactly once in ascending tile order, and the
    shares of a long bin differ by at most one tile."""
    import ctypes
    lib = mi_lib.library().lib
    fn = lib.mi_debug_tile_owner
    fn.restype = ctypes.c_int
    out = (ctypes.c_uint32 * 3)()
    for nt in list(range(1, 71)) + [127, 128, 1000, 4097]:
        for b in (0, 1, 5, 43, 255, 256):
            shares = []
            for x in range(8):
                assert fn(0, nt, b, x, out) == 0
                shares.append(out[2])
            assert sum(shares) == nt
            if nt >= 8:
                assert max(shares) - min(shares) <= 1
            seen = [[] for _ in range(8)]
            for tl in range(nt):
                assert fn(tl, nt, b, 0, out) == 0
                seen[out[0]].append(out[1])
            for x in range(8):
                assert seen[x] == list(range(shares[x])), (nt, b, x)


def test_every_environment_variable_is_read_in_one_place():
    """csrc/knobs.hpp is the only place the physics library reads the environment (once per world, into one typed struct); every MI_* variable the
    tests, the bench and the tools set is one it knows (a misspelt knob would silently test the default path)."""
    import re
    root = Path(__file__).resolve().parent.parent
    csrc = root / "d3d12renderer_amd" / "csrc"
    for f in sorted(csrc.iterdir()):
        if f.name in ("knobs.hpp", "learning.cpp") or f.suffix not in (".hip", ".hpp", ".inc", ".cpp"):
            continue
        assert "getenv" not in f.read_text(), f"{f.name} reads the environment itself"
    known = set(re.findall(r'"(MI_[A-Z0-9_]+)"', (csrc / "knobs.hpp").read_text())) | set(re.findall(r'"(MI_[A-Z0-9_]+)"', (csrc / "learning.cpp").read_text()))
    known |= {"MI_PHYSICS_LIB", "MI_LEARNING_LIB", "MI_SHARD_TRANSPORT"}          # the Python side's own (which build of the library to load; bench.py's default transport)
    known |= set(re.findall(r'"(MI_[A-Z0-9_]+)"', (root / "oracle" / "ora_learning_backend.h").read_text()))   # the checker's backend header for learning.cpp (test side)
    assert "ora_" not in (csrc / "learning.cpp").read_text(), "the product's learning source names the checker's ABI"
    used = set()
    for f in list((root / "tests").glob("*.py")) + list((root / "tools").glob("*.py")) + list((root / "tools").glob("*.sh")) + [root / "bench.py", root / "__graft_entry__.py"]:
        used |= set(re.findall(r'\b(MI_[A-Z0-9_]+)\b', f.read_text()))
    used = {u for u in used if not u.startswith(("MI_ERR", "MI_OK", "MI_EVENT", "MI_CONSTRAINT", "MI_OBJECT", "MI_API", "MI_SEAM", "MI_SHAPE", "MI_COLLIDER", "MI_DBG_TIMELINE", "MI_CLIP", "MI_SHARD_RECORD", "MI_LEARNING_API"))}
    assert used <= known, f"unknown knobs: {sorted(used - known)}"
