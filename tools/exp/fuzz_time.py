"""Development: one fuzz world (tools/gpu_fuzz.py) on the GPU with the wall time of every step and of every action between steps: python tools/exp/fuzz_time.py SEED"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gpu_fuzz                                   # noqa: E402
import d3d12renderer_amd as mi                    # noqa: E402

seed = int(sys.argv[1])
sc, bodies, rng = gpu_fuzz.make_world_description(seed)
events = bool(rng.random() < 0.5)
plan = gpu_fuzz.plan_actions(seed, 40)
w = sc.populate(mi.create_world(0)); s = sc.settings()
if events:
    w.enable_events(True)
alive = list(int(b) for b in bodies)
for i, (mode, u, r, v) in enumerate(plan):
    t0 = time.time()
    if mode[0] == "frame":
        w.step(s, mode[1])
    else:
        w.step_fixed(s, sc.dt, 1)
    c = w.counts(); t1 = time.time()
    if events:
        w.poll_events()
    st = w.get_body_states(np.asarray(alive, np.uint32)) if alive else None
    t2 = time.time()
    alive = gpu_fuzz.apply_action(w, u, r, v, alive, sc); t3 = time.time()
    print(f"step {i:2d} {mode[0]:5s} step {1e3 * (t1 - t0):8.2f} ms  read {1e3 * (t2 - t1):7.2f} ms  action {1e3 * (t3 - t2):7.2f} ms  modes {w.step_mode_stats()}  contacts {c['num_contacts']} colors {c['num_colors']} solver {w.solver_kind()}", flush=True)
