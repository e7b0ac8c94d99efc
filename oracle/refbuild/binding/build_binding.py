#!/usr/bin/env python
"""Recipe for oracle/_ref/binding_check: the reference's own scene + physics sources (read where they lie under /root/reference, patched as
oracle/refbuild/build_ref.py describes, never committed) compiled together with
  * binding/physics_mi355x.cpp   — the backend stub a maintainer of the reference would add, against the reference's REAL headers,
  * binding/drive_reference_scenes.cpp — a driver that builds the reference's demo and ragdoll scenes with the reference's own API and steps them
    once with the reference's physicsStep and once through the stub,
and linked against d3d12renderer_amd/libmi_physics.so.  One more textual patch than build_ref.py applies, and it is the one a maintainer would
make: scene_entity::addComponent (src/scene/scene.h) calls the stub's hook, `miOnPhysicsComponentChanged(registry);`, after adding a component.
Test infrastructure: tests/test_gpu_binding.py runs the binary on the GPU box (it travels there prebuilt, like libref.so)."""
import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle.refbuild import build_ref as R   # noqa: E402

OUT = R.OUT_DIR / "binding_check"
PRODUCT_DIR = ROOT / "d3d12renderer_amd"

HOOK_DECL = "\n// physics backend on an MI355X (src/physics/physics_mi355x.cpp): the physics topology of a registry changed\nvoid miOnPhysicsComponentChanged(entt::registry* registry);\n"


def patch_scene_h(text):
    """The maintainer's one-line hook: after any component was added through scene_entity::addComponent."""
    text = R._insert_after(text, "struct scene_entity\n{", "")          # (anchor check)
    i = text.index("struct scene_entity\n{")
    text = text[:i] + HOOK_DECL + text[i:]
    anchor = "\t\treturn *this;\n\t}\n\n\ttemplate <typename component_t>\n\tbool hasComponent()"
    assert text.count(anchor) == 1
    return text.replace(anchor, "\t\tmiOnPhysicsComponentChanged(registry);\n" + anchor)


def build(force=False, verbose=False, backend="product"):
    """backend "product": linked against d3d12renderer_amd/libmi_physics.so (needs a GPU to run).  backend "oracle": the same stub and driver over the
    CPU oracle's ABI (oracle_backend.cpp forwards the mi_* calls to liboracle.so, canonical order): runs in the CPU test suite."""
    OUT = globals()["OUT"] if backend == "product" else globals()["OUT"].with_name("binding_check_oracle")
    src_root = R.REFERENCE_ROOT / "src"
    if not src_root.exists():
        if OUT.exists():
            return OUT
        raise RuntimeError(f"{src_root} not present and no prebuilt {OUT}")
    lib = PRODUCT_DIR / "libmi_physics.so"
    deps = [src_root / f for f in R.FILES] + [HERE / "physics_mi355x.cpp", HERE / "physics_mi355x.h", HERE / "drive_reference_scenes.cpp", HERE / "oracle_backend.cpp", Path(__file__), Path(R.__file__), R.HERE / "ref_pch.h",
            ROOT / "include" / "mi_physics.h", ROOT / "include" / "mi_constraints.h"] + list((R.HERE / "stubs").rglob("*.h"))
    if not force and OUT.exists() and all(p.stat().st_mtime <= OUT.stat().st_mtime for p in deps if p.exists()):
        return OUT
    R.OUT_DIR.mkdir(exist_ok=True)
    tmp = Path(tempfile.mkdtemp(prefix="bindbuild_"))
    try:
        for rel in R.FILES:
            dst = tmp / "src" / rel
            dst.parent.mkdir(parents=True, exist_ok=True)
            text = R.patch_text(rel, (src_root / rel).read_text(encoding="utf-8", errors="replace"))
            if rel == "scene/scene.h":
                text = patch_scene_h(text)
            dst.write_text(text)
        flags = ["-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-fms-extensions", "-mavx2", "-mfma", "-msse4.1", "-fno-lax-vector-conversions", "-DPHYSICS_ONLY",
                 "-DPHYSICS_BACKEND_MI355X", "-fdelayed-template-parsing", "-w", "-include", str(R.HERE / "ref_pch.h"), "-I", str(R.HERE / "stubs"), "-I", str(tmp / "src"),
                 "-I", str(tmp / "src" / "physics"), "-I", str(R.REFERENCE_ROOT / "ext"), "-I", str(ROOT / "include"), "-I", str(HERE)]
        units = [u for u in R.UNITS if u != "learning/learned_locomotion.cpp"]
        objs = []
        ours = ["physics_mi355x.cpp", "drive_reference_scenes.cpp"] + (["oracle_backend.cpp"] if backend == "oracle" else [])
        for u in units + ours + ["ora_det.cpp"]:
            src = (HERE / u) if u in ours else (R.HERE.parent / u) if u == "ora_det.cpp" else (tmp / "src" / u)
            obj = tmp / (u.replace("/", "_") + ".o")
            cmd = [R.CLANG, *flags, "-c", str(src), "-o", str(obj)]
            if u == "ora_det.cpp":
                cmd = [R.CLANG, "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-c", str(src), "-o", str(obj)]
            if u == "oracle_backend.cpp":   # our own file: no reference prefix header
                cmd = [R.CLANG, "-std=c++17", "-O2", "-fPIC", "-I", str(ROOT / "include"), "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"unit {u} failed to compile")
            objs.append(str(obj))
        # $ORIGIN-relative run path: the binary lies in oracle/_ref/, the library in d3d12renderer_amd/
        link = ["-L", str(PRODUCT_DIR), "-l:libmi_physics.so", "-Wl,-rpath,$ORIGIN/../../d3d12renderer_amd"] if backend == "product" else \
               ["-L", str(ROOT / "oracle" / "_build"), "-l:liboracle.so", "-Wl,-rpath,$ORIGIN/../_build"]
        if backend == "oracle":   # (ora_det.cpp is in liboracle.so as well: the program's own copy wins, same code)
            import oracle
            oracle.build()
        r = subprocess.run([R.CLANG, "-o", str(OUT), *objs, *link, "-lpthread", "-ldl"], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link of binding_check failed")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose="-v" in sys.argv))
    print(build(force=True, verbose="-v" in sys.argv, backend="oracle"))
