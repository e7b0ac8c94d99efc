"""The reference-side binding as a compiled program, on the GPU: oracle/_ref/binding_check = the reference's own scene / physics sources + the backend
stub oracle/refbuild/binding/physics_mi355x.cpp (compiled against the reference's real headers, hook patched into scene_entity::addComponent) linked
against d3d12renderer_amd/libmi_physics.so.  Built where /root/reference exists (oracle/refbuild/binding/build_binding.py, by __graft_entry__.build()),
it travels to the GPU box prebuilt.  See tests/test_reference_pin.py::test_reference_side_binding_compiles_against_the_reference_and_runs for what it checks."""
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_gpu_reference_scene_code_through_the_stub_equals_the_reference(oracle_mod, mi_lib):
    exe = oracle_mod.REF_LIB.with_name("binding_check")
    if not exe.exists():
        if not oracle_mod.reference_available():
            pytest.skip("oracle/_ref/binding_check is not here and /root/reference is not mounted")
        exe, _ = oracle_mod.build_binding()
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900)
    print(r.stdout[-2500:])
    assert r.returncode == 0 and "BINDING CHECK OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if "reference order" in l]
    assert len(lines) == 2 and all("bit-identical steps 2" in l and "max position difference 0 m" in l for l in lines), r.stdout
