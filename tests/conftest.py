import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def mi_lib():
    """The HIP product library; building is allowed on CPU (hipcc cross-compiles), loading too."""
    import d3d12renderer_amd as mi
    from d3d12renderer_amd import build
    build.build()
    try:   # torch bundles its own HIP runtime: initialise it before the library touches the device (as bench.py does), not in between
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:   # noqa: BLE001 - CPU-only machine
        pass
    return mi
