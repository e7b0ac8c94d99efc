"""Every BASELINE configuration at full size, stepped speculatively (the default) and synchronously (MI_ASYNC=0: exact sizes read back inside every step — the path of a world's
first step and of every re-run): the two must end in the same bits.  One line per scene."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
CASES = (("cfg1 4096 spheres", lambda: scenes.sphere_drop(16), 200), ("cfg2 65536 mixed", lambda: scenes.mixed_stack(64, 16, 64), 100), ("cfg3 262144 boxes", lambda: scenes.obb_pile(128, 16, 128), 100),
         ("cfg4 1024 ragdolls", lambda: scenes.ragdolls(32, 32), 150), ("cfg5 256 vehicles", lambda: scenes.vehicles(16, 16), 150), ("terrain 65536", lambda: scenes.terrain_big(), 120),
         ("zones 6912 (triggers, force fields)", lambda: scenes.zones(48, 3, 48), 150), ("pile 1048576", lambda: scenes.obb_pile(256, 16, 256), 60))
ok = True
for name, make, steps in CASES:
    res = {}
    for mode in ("speculative", "synchronous"):
        if mode == "synchronous": os.environ["MI_ASYNC"] = "0"
        else: os.environ.pop("MI_ASYNC", None)
        sc = make(); w = sc.populate(mi.create_world(0)); t = time.time()
        w.step_fixed(sc.settings(), sc.dt, steps)
        p, q = w.physics_transforms(); v, a = w.velocities()
        res[mode] = (hashlib.sha1(p.tobytes() + q.tobytes() + v.tobytes() + a.tobytes()).hexdigest()[:16], w.counts()["num_contacts"], w.step_mode_stats(), round(time.time() - t, 2))
        w.close()
    same = res["speculative"][:2] == res["synchronous"][:2]; ok = ok and same
    print(f"[sync-vs-spec] {name}, {steps} steps: {'identical' if same else 'DIFFERENT'} {res}", flush=True)
os.environ.pop("MI_ASYNC", None)
print("ALL IDENTICAL" if ok else "MISMATCH")
