"""CPU model of "chained slots" (VERDICT r05 item 3b): a lane solves TWO manifolds that share a body back to back with that body in registers, so that body's hand-over
between them costs no remote hop.  On the canonical schedule of a settled box pile (oracle, canonical order) the dependency graph of 20 sweeps is walked with
    t(m) = max over its dynamic bodies b of ( t(previous manifold of b) + hop(edge) ) + solve,      hop = 1 for a remote hand-over, CHAIN for a chained one,
chains chosen greedily: bodies in descending degree, consecutive colours of a body paired while both manifolds are still free (a manifold can receive at most one body
in registers and pass on at most one).  Printed: the critical path in units of a remote hop, unchained against chained, for several solve / chain costs.
python tools/chained_slots_model.py 32 8 32 600"""
import sys, ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import oracle
from d3d12renderer_amd import scenes
nx, ny, nz, steps = (int(v) for v in sys.argv[1:5])
sc = scenes.obb_pile(nx, ny, nz)
w = sc.populate(oracle.create_world(oracle.ORDER_CANONICAL)); s = sc.settings(); NB = sc.num_bodies
w.step_fixed(s, sc.dt, steps)
con = w.contacts()
keep = np.ones(len(con), bool); cab = np.stack([con["collider_a"], con["collider_b"]], axis=1); keep[1:] = (cab[1:] != cab[:-1]).any(axis=1)
ba = con["body_a"][keep].astype(np.int64); bb = con["body_b"][keep].astype(np.int64); nm = len(ba)
col = np.zeros(nm, np.uint32); w.L.fn("world_get_manifold_colors")(w.h, col.ctypes.data_as(C.c_void_p), C.c_uint32(nm))
ncol = int(col.max()) + 1
order = np.argsort(col, kind="stable")
# per body: its manifolds in colour order
per_body = [[] for _ in range(NB)]
for m in order:
    for b in (int(ba[m]), int(bb[m])):
        if b < NB: per_body[b].append(int(m))
deg = np.array([len(x) for x in per_body])
# greedy chains: (m1 -> m2 through body b): m1 passes b on in registers
passes = {}; receives = {}
for b in np.argsort(-deg, kind="stable"):
    ms = per_body[b]
    i = 0
    while i + 1 < len(ms):
        m1, m2 = ms[i], ms[i + 1]
        if m1 not in passes and m2 not in receives:
            passes[m1] = (m2, int(b)); receives[m2] = (m1, int(b)); i += 2
        else:
            i += 1
chained_edges = len(passes)
total_edges = int(sum(max(0, d - 1) for d in deg))
print(f"{NB} bodies, {nm} manifolds, {ncol} colours, max degree {int(deg.max())}; intra-sweep hand-overs {total_edges}, of which chained {chained_edges} ({chained_edges / max(1, total_edges):.0%})")


def critical(chain_cost, solve, sweeps=20, use_chains=True):
    last = np.zeros(NB + 1); last_m = -np.ones(NB + 1, np.int64)
    for _ in range(sweeps):
        for m in order:
            m = int(m); t = 0.0
            for b in (int(ba[m]), int(bb[m])):
                if b >= NB: continue
                hop = chain_cost if (use_chains and last_m[b] >= 0 and receives.get(m, (None, None)) == (int(last_m[b]), b)) else 1.0
                t = max(t, last[b] + (hop if last_m[b] >= 0 else 0.0))
            t += solve
            for b in (int(ba[m]), int(bb[m])):
                if b < NB: last[b] = t; last_m[b] = m
    return float(last.max())


for solve, chain in ((0.0, 0.0), (0.17, 0.0), (0.17, 0.1)):
    a = critical(chain, solve, use_chains=False); b = critical(chain, solve, use_chains=True)
    print(f"solve {solve:.2f}, chained hop {chain:.2f} (units of one remote hop): critical path of 20 sweeps {a:.1f} -> {b:.1f} ({b / a:.2f} x)")
