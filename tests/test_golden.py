"""The oracle, rebuilt from source on this machine, must reproduce the committed vectors bit for bit (tests/golden/*.npz, generated
by tools/make_golden.py): `*_reference.npz` are outputs of the REFERENCE ITSELF (its sources compiled into oracle/_ref and stepped
through its own physicsStep), `*_canonical.npz` the oracle's replay of the GPU schedule."""
import sys
from pathlib import Path
import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import make_golden  # noqa: E402


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
@pytest.mark.parametrize("tag", ["canonical", "reference"])
def test_oracle_reproduces_golden(oracle_mod, name, tag):
    make, steps = make_golden.CASES[name]
    order = oracle_mod.ORDER_CANONICAL if tag == "canonical" else oracle_mod.ORDER_REFERENCE
    got = make_golden.run_case(make, steps, order)
    want = np.load(ROOT / "tests" / "golden" / f"{name}_{tag}.npz")
    assert np.array_equal(got["counts"], want["counts"])
    for k in ("pos", "rot", "lin", "ang"):
        assert got[k].tobytes() == want[k].tobytes(), k
    if tag == "reference":
        assert "reference" in str(want["source"]), "the *_reference vectors must come from the reference build, not from the oracle"
        if oracle_mod.reference_available():      # and the reference build on this machine still produces them
            ref = make_golden.run_case(make, steps, order, from_reference=True)
            assert np.array_equal(ref["counts"], want["counts"]) and all(ref[k].tobytes() == want[k].tobytes() for k in ("pos", "rot", "lin", "ang"))
