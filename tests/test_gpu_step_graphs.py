"""Step graphs (csrc/launcher.hpp, world.hip): a speculative step of a small scene whose enqueued work — every kernel, grid, pointer,
size and scalar — is bit-identical to that of a captured step is replayed with ONE hipGraphLaunch.  Replays must be invisible in
the results.  The check runs in a subprocess WITHOUT torch, so that the library binds the system HIP runtime (7.2), where the
graphs are enabled; in this process (torch's 7.0.x runtime is loaded first) the library keeps them off by default ("Step graphs")."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_step_graph_replays_are_bit_exact(mi_lib, record_property):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "check_step_graphs.py")], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    record_property("step_graphs", json.dumps(out)); print(out)
    replayed = 0
    for name, res in out["scenes"].items():
        assert res["first_mismatch"] is None, f"{name}: the graph-replaying world left the plain one at step {res['first_mismatch']}"
        assert res["times_ok"], f"{name}: step / solve timing must survive a replay"
        assert res["plain_stats"][1] == 0
        replayed += res["graph_stats"][1]
        if res["graph_stats"][0]:
            assert res["graph_stats"][1] > res["steps"] // 3, f"{name}: a steady scene should mostly replay ({res['graph_stats']})"
    if not replayed:
        pytest.skip("step graphs are disabled under this HIP runtime")


def test_step_graphs_are_off_under_the_bundled_runtime(mi_lib):
    """In THIS process torch's HIP runtime (7.0.x) is the one the library is bound to: graphs stay off (see module docstring)."""
    import ctypes, os
    if os.environ.get("MI_GRAPH"):
        pytest.skip("MI_GRAPH overrides the default")
    ver = ctypes.c_int(0)
    ctypes.CDLL("libamdhip64.so").hipRuntimeGetVersion(ctypes.byref(ver))
    from d3d12renderer_amd import scenes
    sc = scenes.sphere_drop(6)
    w = sc.populate(mi_lib.create_world(0))
    w.step_fixed(sc.settings(), sc.dt, 40)
    enabled, hits, _c, _p = w.step_graph_stats()
    assert bool(enabled) == (ver.value >= 70200000)
    if not enabled:
        assert hits == 0
