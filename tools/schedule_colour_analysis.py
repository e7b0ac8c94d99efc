"""CPU (oracle, canonical order): how many colours the schedule has against the maximal body degree and against fresh greedy colourings of the same
manifold set, and the longest dependency path of the per-body update sequences over 20 sweeps.  python tools/schedule_colour_analysis.py 32 8 32 400"""
import sys, time, ctypes as C, collections
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle
from d3d12renderer_amd import scenes
nx, ny, nz, steps = (int(v) for v in sys.argv[1:5])
sc = scenes.obb_pile(nx, ny, nz)
w = sc.populate(oracle.create_world(oracle.ORDER_CANONICAL)); s = sc.settings(); NB = sc.num_bodies
def analyse(tag):
    con = w.contacts()
    keep = np.ones(len(con), bool); cab = np.stack([con["collider_a"], con["collider_b"]], axis=1); keep[1:] = (cab[1:] != cab[:-1]).any(axis=1)
    ba = con["body_a"][keep].astype(np.int64); bb = con["body_b"][keep].astype(np.int64); nm = len(ba)
    col = np.zeros(nm, np.uint32)
    rc = w.L.fn("world_get_manifold_colors")(w.h, col.ctypes.data_as(C.c_void_p), C.c_uint32(nm)); assert rc == 0
    deg = np.bincount(np.concatenate([ba[ba < NB], bb[bb < NB]]), minlength=NB)
    hist = np.bincount(col, minlength=1)
    # fresh greedy colourings of the same edge set
    def greedy(order):
        used = collections.defaultdict(int); out = np.zeros(nm, np.int64)
        for m in order:
            a, b = int(ba[m]), int(bb[m]); mask = (used[a] if a < NB else 0) | (used[b] if b < NB else 0)
            c = (~mask & (mask + 1)).bit_length() - 1
            out[m] = c
            if a < NB: used[a] |= 1 << c
            if b < NB: used[b] |= 1 << c
        return out
    rng = np.random.default_rng(1)
    fresh_rand = greedy(rng.permutation(nm))
    dsum = np.where(ba < NB, deg[np.minimum(ba, NB - 1)], 0) + np.where(bb < NB, deg[np.minimum(bb, NB - 1)], 0)
    dmax = np.maximum(np.where(ba < NB, deg[np.minimum(ba, NB - 1)], 0), np.where(bb < NB, deg[np.minimum(bb, NB - 1)], 0))
    fresh_deg = greedy(np.lexsort((rng.random(nm), -dsum)))
    fresh_dmax = greedy(np.lexsort((rng.random(nm), -dsum, -dmax)))
    print(tag, "manifolds", nm, "max degree", int(deg.max()), "deg hist", np.bincount(deg).tolist(), "\n   history colours", int(col.max()) + 1, hist.tolist(),
          "\n   fresh random-order greedy", int(fresh_rand.max()) + 1, np.bincount(fresh_rand).tolist(),
          "\n   fresh degree-sum-first", int(fresh_deg.max()) + 1, np.bincount(fresh_deg).tolist(),
          "\n   fresh max-degree-first", int(fresh_dmax.max()) + 1, np.bincount(fresh_dmax).tolist(), flush=True)
t = time.time()
for i in range(steps):
    w.step_fixed(s, sc.dt, 1)
    if i in (119, 239, 399, 599, 899, 1199, steps - 1): analyse(f"step {i + 1} ({time.time() - t:.0f} s)")

def critical():
    con = w.contacts()
    keep = np.ones(len(con), bool); cab = np.stack([con["collider_a"], con["collider_b"]], axis=1); keep[1:] = (cab[1:] != cab[:-1]).any(axis=1)
    ba = con["body_a"][keep].astype(np.int64); bb = con["body_b"][keep].astype(np.int64); nm = len(ba)
    col = np.zeros(nm, np.uint32); w.L.fn("world_get_manifold_colors")(w.h, col.ctypes.data_as(C.c_void_p), C.c_uint32(nm))
    order = np.argsort(col, kind="stable")
    last = np.zeros(NB + 1, np.int64)
    ncol = int(col.max()) + 1
    for sweep in range(20):
        for m in order:
            a, b = int(ba[m]), int(bb[m])
            t = 1 + max(last[a] if a < NB else 0, last[b] if b < NB else 0)
            if a < NB: last[a] = t
            if b < NB: last[b] = t
        if sweep in (0, 1, 4, 9, 19): print("  after sweep", sweep + 1, "longest dependency path", int(last.max()), "hops; colours x sweeps", ncol * (sweep + 1), "; mean body time", round(float(last[:NB].mean()), 1))
critical()
