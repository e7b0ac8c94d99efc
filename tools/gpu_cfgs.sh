#!/bin/bash
# dev helper: step timing of the other BASELINE configs (parity-test cases, not bench lines) at full size
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/cfgs.py <<'PY'
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
out = {}
for name, make, warm, steps in (("cfg1_spheres_4096", lambda: scenes.sphere_drop(16), 240, 60), ("cfg2_mixed_65536", lambda: scenes.mixed_stack(64, 16, 64), 240, 60),
                                ("cfg4_ragdolls_1024", lambda: scenes.ragdolls(32, 32), 240, 60), ("cfg5_vehicles_256", lambda: scenes.vehicles(16, 16), 240, 60),
                                ("terrain_65536", lambda: scenes.terrain_big(), 300, 60), ("zones_6912", lambda: scenes.zones(48, 3, 48), 200, 60),
                                ("cfg3_settled_pile_262144", lambda: scenes.obb_pile(128, 16, 128), 1500, 60), ("pile_1048576", lambda: scenes.obb_pile(256, 16, 256), 240, 30)):
    import os
    if os.environ.get('CFGS') and not any(k in name for k in os.environ['CFGS'].split(',')): continue
    sc = make()
    w = sc.populate(mi.create_world(0))
    s = sc.settings()
    w.step_fixed(s, sc.dt, warm)
    w.counts(); t0 = time.perf_counter()
    for _ in range(steps): w.step_fixed(s, sc.dt, 1)          # untimed by the library (its default): what a caller gets
    w.counts(); dt = (time.perf_counter() - t0) / steps
    w.set_stage_timing(1); acc = {}                           # then the per-stage breakdown of a few more steps (an event pair per stage: slower)
    for _ in range(10):
        w.step_fixed(s, sc.dt, 1)
        for k, v in w.stage_times().items(): acc[k] = acc.get(k, 0.0) + v / 10
    p, q = w.physics_transforms()
    out[name] = dict(bodies=sc.num_bodies, solver=w.solver_kernel(), solver_kind=w.solver_kind(), ms_per_step=dt * 1e3, steps_per_s=1 / dt, counts=w.counts(), finite=bool(np.isfinite(p).all()), stage_ms={k: round(v, 4) for k, v in acc.items()})
    print(name, json.dumps(out[name]), flush=True)
json.dump(out, open("gpurun_out/cfgs.json", "w"), indent=1)
PY
timeout 900 python /tmp/cfgs.py 2>&1 | tail -8
