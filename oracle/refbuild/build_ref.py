#!/usr/bin/env python
"""Recipe for oracle/_ref/libref.so: the REFERENCE's own physics routines compiled on Linux.

Test infrastructure only (tests/ use it to pin oracle/ against the original code).  Nothing of the reference is
committed: the sources are read where they lie under /root/reference (REFERENCE_ROOT), a handful of textual patches
(listed in PATCHES below, each with its reason) is applied to temporary copies in a scratch directory, the copies
are compiled together with oracle/refbuild/ref_shim.cpp (our extern "C" entry points over the reference's functions)
and only the resulting shared library lands in oracle/_ref/.  The scratch directory is deleted afterwards.

The reference is Windows/MSVC-only code; what the patches change is spelling, never arithmetic:
  * `__m128::m128_f32[i]` / `__m128i::m128i_i32[i]` (MSVC union members of the vector types) -> lane pointers;
  * `_mm_div_epi32` / `_mm256_div_epi32` (Intel SVML, not in clang/gcc) -> lane-wise integer division (unused by the
    routines we call);
  * MSVC accepts a member name declared in two anonymous structs of one union (`vec3::z`, `vec3::x`, `vec4::w`, `vec4::x`);
    the later duplicates are renamed (they alias the same storage and are not referenced by name in the physics code);
  * overload sets that are ambiguous outside MSVC (`reinterpret(__m256i)` …) get explicit casts;
  * headers that drag in EnTT / D3D12 / the job system are replaced by small stand-ins in oracle/refbuild/stubs/.
Compiled with clang -O2 -ffp-contract=off -fno-fast-math (the reference builds with /fp:fast; see DESIGN.md §2 on why the
strict build is the one the oracle is pinned to).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT_DIR = HERE.parent / "_ref"
OUT = OUT_DIR / "libref.so"
OUT_FAST = OUT_DIR / "libref_fast.so"     # timing build: the reference's own flags (premake5.lua:266-267: AVX2 + /fp:fast), libm left alone
REFERENCE_ROOT = Path(os.environ.get("REFERENCE_ROOT", "/root/reference"))
CLANG = os.environ.get("REF_CXX", "/opt/rocm/lib/llvm/bin/clang++")

# reference files that are compiled (translation units) or included by them
FILES = [
    "core/math.h", "core/math.cpp", "core/simd.h", "core/math_simd.h", "core/soa.h", "core/random.h",
    "physics/bounding_volumes.h", "physics/bounding_volumes.cpp", "physics/bounding_volumes_simd.h",
    "physics/collision_gjk.h", "physics/collision_gjk.cpp",
    "physics/collision_epa.h", "physics/collision_epa.cpp",
    "physics/collision_broad.h", "physics/collision_broad.cpp", "physics/collision_narrow.h", "physics/collision_narrow.cpp",
    "physics/collision_sat.h", "physics/constraints.h", "physics/constraints.cpp", "physics/physics.h", "physics/physics.cpp",
    "physics/cloth.h", "physics/cloth.cpp", "physics/rigid_body.h", "physics/rigid_body.cpp",
    "physics/heightmap_collision.h", "physics/heightmap_collision.cpp",
    "core/memory.h", "core/memory.cpp", "core/reflect.h", "core/preprocessor_for_each.h",
    "scene/scene.h", "scene/scene.cpp", "scene/components.h",
    "terrain/heightmap_collider.h", "terrain/heightmap_collider.cpp",
    "physics/ragdoll.h", "physics/ragdoll.cpp", "learning/learned_locomotion.h", "learning/learned_locomotion.cpp",
]
UNITS = ["core/math.cpp", "core/memory.cpp", "physics/bounding_volumes.cpp", "physics/collision_gjk.cpp", "physics/collision_epa.cpp",
         "physics/collision_broad.cpp", "physics/collision_narrow.cpp", "physics/constraints.cpp", "physics/rigid_body.cpp",
         "physics/heightmap_collision.cpp", "physics/cloth.cpp", "physics/physics.cpp", "scene/scene.cpp", "terrain/heightmap_collider.cpp",
         "physics/ragdoll.cpp", "learning/learned_locomotion.cpp"]

LANE = r"([A-Za-z_0-9\.>\-]+)"
# (file glob or None for all, regex, replacement, reason)
PATCHES = [
    (None, LANE + r"\.m(?:128|256|512)_f32\[", r"REF_LANEF(\1)[", "MSVC vector-type union member"),
    (None, LANE + r"\.m(?:128|256|512)i_i32\[", r"REF_LANEI(\1)[", "MSVC vector-type union member"),
    ("core/simd.h", r"_mm_div_epi32\(a, b\)", "ref_div_epi32_128(a, b)", "SVML intrinsic"),
    ("core/simd.h", r"_mm256_div_epi32\(a, b\)", "ref_div_epi32_256(a, b)", "SVML intrinsic"),
    ("physics/collision_narrow.cpp", r"auto& \[outContact, outBodyPair\] = writeContext\.pushContact\(\);",
     "auto&& [outContact, outBodyPair] = writeContext.pushContact();", "MSVC binds a non-const lvalue reference to a temporary pair of references"),
    ("core/simd.h", r"_mm256_loadu_epi32\(i_\)", "_mm256_loadu_si256((const __m256i*)i_)", "AVX512VL-gated spelling of an unaligned 256-bit load (same vmovdqu)"),
    ("core/simd.h", r"_mm256_storeu_epi32\(i_, i\)", "_mm256_storeu_si256((__m256i*)i_, i)", "same, store"),
    ("core/simd.h", r"_mm512_div_epi32\(a, b\)", "ref_div_epi32_512(a, b)", "SVML intrinsic"),
]


def patch_text(rel, text):
    for glob, pat, rep, _why in PATCHES:
        if glob is None or glob == rel:
            text = re.sub(pat, rep, text)
    fn = SPECIAL.get(rel)
    return fn(text) if fn else text


def _math_h(text):
    # duplicate member names in the anonymous structs of vec3 / vec4 (accepted by MSVC only)
    text = text.replace("\t\tvec2 xy;\n\t\tfloat z;\n", "\t\tvec2 xy;\n\t\tfloat z_dup;\n", 1)
    text = text.replace("\t\tfloat x;\n\t\tvec2 yz;\n", "\t\tfloat x_dup;\n\t\tvec2 yz;\n", 1)
    text = text.replace("\t\tvec3 xyz;\n\t\tfloat w;\n", "\t\tvec3 xyz;\n\t\tfloat w_dup;\n", 1)
    text = text.replace("\t\tfloat x;\n\t\tvec3 yzw;\n", "\t\tfloat x_dup;\n\t\tvec3 yzw;\n", 1)
    return text


def _math_simd_h(text):
    # same duplicate-member pattern in the wide vector unions, and class-template names used without arguments where no
    # deduction is possible (`wN_quat result;` — MSVC resolves them inside the template to the enclosing specialisation)
    text = text.replace("\t\twN_vec2<simd_t> xy;\n\t\tsimd_t z;\n", "\t\twN_vec2<simd_t> xy;\n\t\tsimd_t z_dup;\n", 1)
    text = text.replace("\t\twN_vec3<simd_t> xyz;\n\t\tsimd_t w;\n", "\t\twN_vec3<simd_t> xyz;\n\t\tsimd_t w_dup;\n", 1)
    text = re.sub(r"\b(wN_(?:quat|vec2|vec3|vec4|mat2|mat3|mat4)) (result|p)([;(])", r"\1<simd_t> \2\3", text)
    return text


def _scene_cpp(text):
    # cloneTo / copyEntity name renderer components (mesh, raytrace, animation) even under PHYSICS_ONLY; the editor-only
    # functions are dropped, the constructor (the three owning groups), clearAll and deleteEntity stay
    a = text.index("void game_scene::cloneTo")
    b = text.index("void game_scene::deleteEntity")
    return text[:a] + text[b:]


def _insert_after(text, anchor, addition):
    i = text.index(anchor) + len(anchor)
    return text[:i] + addition + text[i:]


def _physics_cpp(text):
    # read-only instrumentation: hand the step's temporaries to ref_shim.cpp before the arena is rewound, and let the shim
    # register hull geometry (allocateBoundingHullGeometry loads a mesh FILE and is compiled out under PHYSICS_ONLY)
    decl = ("\nvoid ref_tap_broadphase(const collider_pair* pairs, uint32 numPairs);\n"
            "void ref_tap_step(uint32 numRigidBodies, uint32 numColliders, const bounding_box* aabbs, const collider_union* worldSpaceColliders,\n"
            "\tuint32 numBroadphaseOverlaps, uint32 numCollisions, uint32 numContacts, const collision_contact* contacts, const constraint_body_pair* bodyPairs,\n"
            "\tconst collider_pair* collidingPairs, const uint8* contactCountPerCollision);\n"
            "uint32 ref_allocate_hull_geometry(vec3* vertices, uint32 numVertices, indexed_triangle16* triangles, uint32 numTriangles)\n"
            "{\n\tuint32 index = (uint32)boundingHullGeometries.size();\n"
            "\tboundingHullGeometries.push_back(bounding_hull_geometry::fromMesh(vertices, numVertices, triangles, numTriangles));\n\treturn index;\n}\n")
    text = _insert_after(text, "static std::vector<bounding_hull_geometry> boundingHullGeometries;\n", decl)
    text = _insert_after(text, "uint32 numBroadphaseOverlaps = broadphase(scene, worldSpaceAABBs, arena, overlappingColliderPairs, settings.simdBroadPhase);\n",
                         "\tref_tap_broadphase(overlappingColliderPairs, numBroadphaseOverlaps);\n")
    a = text.index("static void physicsStepInternal(")
    b = text.index("\tarena.resetToMarker(marker);\n}", a)
    tap = ("\tref_tap_step(numRigidBodies, numColliders, worldSpaceAABBs, worldSpaceColliders, numBroadphaseOverlaps, narrowPhaseResult.numCollisions,\n"
           "\t\tnarrowPhaseResult.numContacts, contacts, collisionBodyPairs, collidingColliderPairs, contactCountPerCollision);\n")
    return text[:b] + tap + text[b:]


def _collision_broad_cpp(text):
    return text + ("\n// instrumentation (oracle/refbuild): the axis the next sweep will sort along\n"
                   "uint32 ref_sap_sorting_axis(game_scene& scene)\n{\n\tsap_context* c = tryGetContextVariable<sap_context>(scene.registry);\n"
                   "\treturn c ? c->sortingAxis : 0;\n}\n")


def _learned_locomotion_cpp(text):
    # test hooks: a settable push-RNG state instead of time(0), and the choice of the scalar step (the DLL steps with the
    # default physics_settings, i.e. the AVX2 path, which is not bit-comparable with anything)
    text = text.replace("static random_number_generator rng = { (uint32)time(0) };",
                        "static random_number_generator rng = { (uint32)time(0) };\nstatic bool refLearningSimd = true;\n"
                        "extern \"C\" void refLearningConfigure(unsigned long long rngState, int simd) { rng.state = rngState; refLearningSimd = simd != 0; }")
    text = text.replace("\tphysicsSettings.frameRate = 60;\n",
                        "\tphysicsSettings.frameRate = 60;\n\tphysicsSettings.simdBroadPhase = physicsSettings.simdNarrowPhase = physicsSettings.simdConstraintSolver = refLearningSimd;\n", 1)
    return text


SPECIAL = {"learning/learned_locomotion.cpp": _learned_locomotion_cpp, "core/math.h": _math_h, "core/math_simd.h": _math_simd_h, "scene/scene.cpp": _scene_cpp,
           "physics/physics.cpp": _physics_cpp, "physics/collision_broad.cpp": _collision_broad_cpp}


def build(force=False, verbose=False, keep=False, variant="strict"):
    """variant "strict": -O2, IEEE float evaluation, transcendental calls routed to ora_det.cpp — what the oracle is pinned to.
    variant "fast": -O2 -ffast-math -mavx2 -mfma (MSVC /O2 /fp:fast /arch:AVX2; clang -O3 miscompiles or exposes UB here and crashes) with the C library's own libm — the reference as its project file builds it,
    used only as the timed CPU baseline (bench.py cpu_baseline kind "reference")."""
    OUT = OUT_FAST if variant == "fast" else globals()["OUT"]
    src_root = REFERENCE_ROOT / "src"
    if not src_root.exists():
        if OUT.exists():
            return OUT          # prebuilt library travels to the GPU box; the reference does not
        raise RuntimeError(f"{src_root} not present and no prebuilt {OUT}")
    deps = [src_root / f for f in FILES] + [HERE / "ref_shim.cpp", HERE / "ref_pch.h", HERE.parent / "ora_det.cpp", HERE.parent / "ora_math.h", Path(__file__)] + list((HERE / "stubs").rglob("*.h"))
    if not force and OUT.exists() and all(p.stat().st_mtime <= OUT.stat().st_mtime for p in deps if p.exists()):
        return OUT
    OUT_DIR.mkdir(exist_ok=True)
    tmp = Path(tempfile.mkdtemp(prefix="refbuild_"))
    try:
        for rel in FILES:
            dst = tmp / "src" / rel
            dst.parent.mkdir(parents=True, exist_ok=True)
            dst.write_text(patch_text(rel, (src_root / rel).read_text(encoding="utf-8", errors="replace")))
        opt = os.environ.get("REF_FAST_FLAGS", "-O2 -ffast-math").split() + ["-DREF_NATIVE_LIBM"] if variant == "fast" else ["-O2", "-ffp-contract=off", "-fno-fast-math"]
        flags = ["-std=c++17", *opt, "-fPIC", "-fms-extensions", "-mavx2", "-mfma", "-msse4.1",
                 "-fno-lax-vector-conversions", "-DPHYSICS_ONLY", "-fdelayed-template-parsing", "-w", "-include", str(HERE / "ref_pch.h"), "-I", str(HERE / "stubs"), "-I", str(tmp / "src"), "-I", str(tmp / "src" / "physics"), "-I", str(REFERENCE_ROOT / "ext"), "-I", str(HERE.parent.parent / "include")]
        objs = []
        for u in UNITS + ["ref_shim.cpp", "ora_det.cpp"]:
            src = (HERE / u) if u == "ref_shim.cpp" else (HERE.parent / u) if u == "ora_det.cpp" else (tmp / "src" / u)
            obj = tmp / (u.replace("/", "_") + ".o")
            cmd = [CLANG, *flags, "-c", str(src), "-o", str(obj)]
            if u == "learning/learned_locomotion.cpp":   # the environment's own state / reward arithmetic (host code in the product too) keeps the C library's acos / exp
                cmd.insert(1, "-DREF_NATIVE_LIBM")
            if u == "ora_det.cpp":      # our own file: no reference prefix header
                cmd = [CLANG, "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"reference unit {u} failed to compile")
            objs.append(str(obj))
        r = subprocess.run([CLANG, "-shared", "-o", str(OUT), *objs], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link of libref.so failed")
    finally:
        if keep:
            print("scratch kept:", tmp)
        else:
            shutil.rmtree(tmp, ignore_errors=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose="-v" in sys.argv, keep="--keep" in sys.argv))
    print(build(force=True, verbose="-v" in sys.argv, variant="fast"))
