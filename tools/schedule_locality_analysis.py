"""CPU (oracle, canonical order): if every compute unit owned a contiguous spatial share of every (colour, contacts) bin and hand-overs between manifolds of ONE unit were cheaper
(LDS instead of L2), how much shorter would the solver's critical path be?  python tools/schedule_locality_analysis.py 32 8 32 400 <units> <local hop us> <remote hop us>"""
import sys, ctypes as C
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle
from d3d12renderer_amd import scenes
nx, ny, nz, steps, ncu = (int(v) for v in sys.argv[1:6])
sc = scenes.obb_pile(nx, ny, nz)
w = sc.populate(oracle.create_world(oracle.ORDER_CANONICAL)); s = sc.settings(); NB = sc.num_bodies
w.step_fixed(s, sc.dt, steps)
con = w.contacts()
keep = np.ones(len(con), bool); cab = np.stack([con["collider_a"], con["collider_b"]], axis=1); keep[1:] = (cab[1:] != cab[:-1]).any(axis=1)
first = np.flatnonzero(keep); cnt = np.diff(np.append(first, len(con)))
ba = con["body_a"][keep].astype(np.int64); bb = con["body_b"][keep].astype(np.int64); nm = len(ba)
col = np.zeros(nm, np.uint32); w.L.fn("world_get_manifold_colors")(w.h, col.ctypes.data_as(C.c_void_p), C.c_uint32(nm))
pt = con["point"][first]   # a contact point of the manifold: its place
def morton(ix, iz):
    c = np.zeros(len(ix), np.int64)
    for b in range(10): c |= ((ix >> b) & 1) << (2 * b) | ((iz >> b) & 1) << (2 * b + 1)
    return c
mn = pt.min(axis=0); mx = pt.max(axis=0)
ix = ((pt[:, 0] - mn[0]) / (mx[0] - mn[0] + 1e-6) * 63).astype(np.int64); iz = ((pt[:, 2] - mn[2]) / (mx[2] - mn[2] + 1e-6) * 63).astype(np.int64)
key = morton(ix, iz)
# tiles: per (colour, count) bin, slots in spatial order, 64 per tile; a tile's owner: its relative position in the bin -> CU (contiguous ownership)
owner = np.zeros(nm, np.int64)
for c in range(int(col.max()) + 1):
    for k in range(1, 5):
        idx = np.flatnonzero((col == c) & (cnt == k))
        if not len(idx): continue
        idx = idx[np.argsort(key[idx], kind="stable")]
        nt = (len(idx) + 63) // 64
        tile = np.arange(len(idx)) // 64
        owner[idx] = np.minimum(ncu - 1, (tile * ncu) // nt)
order = np.argsort(col, kind="stable")
lastOwner = -np.ones(NB + 1, np.int64); lastT = np.zeros(NB + 1, np.int64)
local = remote = 0
LAT_L, LAT_R = float(sys.argv[6]), float(sys.argv[7])   # hop cost when both dependencies are CU-local / otherwise
t_body = np.zeros(NB + 1)
for sweep in range(20):
    for m in order:
        a, b = int(ba[m]), int(bb[m]); o = owner[m]
        deps = [x for x in (a, b) if x < NB]
        loc = all(lastOwner[x] in (-1, o) for x in deps)
        if sweep > 0:
            if loc: local += 1
            else: remote += 1
        t = max([t_body[x] for x in deps] + [0.0]) + (LAT_L if loc else LAT_R)
        for x in deps: t_body[x] = t; lastOwner[x] = o
print(f"manifolds {nm}, colours {int(col.max()) + 1}, CUs {ncu}: hand-overs with both producers on the consumer's CU {local / (local + remote):.3f}; critical path {t_body.max():.0f} us with hops {LAT_L} / {LAT_R} us (all remote: {20 * (int(col.max()) + 1) * LAT_R:.0f})")
