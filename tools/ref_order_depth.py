"""How deep the dependency DAG of the per-body update sequences is in the REFERENCE's constraint order (CPU, libref.so; needs /root/reference):
level(m) = 1 + max(level of the previous manifold on body A, on body B), static bodies excluded.  The dataflow kernels schedule 64 levels (colours)."""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle
from d3d12renderer_amd import scenes
from helpers import manifold_order
def depth(con):
    ab = manifold_order(con)
    # bodies of a manifold: contact records carry body ids? use body_a/body_b fields if present
    names = con.dtype.names
    keep = np.ones(len(con), bool); cab = np.stack([con["collider_a"], con["collider_b"]], axis=1); keep[1:] = (cab[1:] != cab[:-1]).any(axis=1)
    ba = con["body_a"][keep]; bb = con["body_b"][keep]
    last = {}
    mx = 0; hist = {}
    for a, b in zip(ba.tolist(), bb.tolist()):
        l = 0
        if a < NB and a in last: l = max(l, last[a])
        if b < NB and b in last: l = max(l, last[b])
        l += 1
        if a < NB: last[a] = l
        if b < NB: last[b] = l
        mx = max(mx, l)
    return mx, len(ba)
which = sys.argv[1]
sc = {"cfg3": lambda: scenes.obb_pile(32, 16, 32), "cfg2": lambda: scenes.mixed_stack(32, 16, 32), "cfg1": lambda: scenes.sphere_rain(4096) if hasattr(scenes, "sphere_rain") else None}[which]()
NB = sc.num_bodies
ref = sc.populate(oracle.create_reference_world()); s = sc.settings()
print(ref.contacts().dtype.names)
t = time.time()
for i in range(int(sys.argv[2])):
    ref.step_fixed(s, sc.dt, 1)
    if i % 20 == 19:
        con = ref.contacts(); print(i, depth(con), ref.counts()["num_contacts"], round(time.time() - t, 1), flush=True)
