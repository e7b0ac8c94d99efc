ulimit -c 0
cat > /tmp/np.py <<'PY'
import sys, os
sys.path.insert(0, ".")
os.environ["MI_PHYSICS_LIB"] = "d3d12renderer_amd/libmi_physics_sat.so"
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.obb_pile(128, 16, 128, solver_iterations=20)
w = sc.populate(mi.create_world(0))
w.step_fixed(sc.settings(), sc.dt, 250)
for flag in ("0", "1"):
    os.environ["MI_SATONLY"] = flag
    w.step_fixed(sc.settings(), sc.dt, 1)
    print("satonly", flag, {k: round(v, 3) for k, v in w.stage_times().items()}, w.counts()["num_collisions"], w.counts()["num_contacts"])
PY
timeout 100 python /tmp/np.py 2>&1 | tail -2
