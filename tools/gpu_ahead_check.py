import sys, time
sys.path.insert(0, ".")
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.obb_pile(128, 16, 128); s = sc.settings()
w = sc.populate(mi.create_world(0))
w.step_fixed(s, sc.dt, 240)
print("after settle (one call of 240 steps):", w.debug_step_ahead_stats())
w.counts(); t0 = time.perf_counter()
for _ in range(60): w.step_fixed(s, sc.dt, 1)
w.counts(); print("60 single steps: %.4f ms/step" % ((time.perf_counter() - t0) / 60 * 1e3), w.debug_step_ahead_stats())
w.set_stage_timing(3)
w.counts(); t0 = time.perf_counter()
for _ in range(60): w.step_fixed(s, sc.dt, 1)
w.counts(); print("60 single steps, timing level 3: %.4f ms/step" % ((time.perf_counter() - t0) / 60 * 1e3), w.debug_step_ahead_stats())
