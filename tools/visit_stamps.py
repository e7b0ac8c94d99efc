"""Per-visit wall-clock stamps of k_contact_solve_persist (a -DMI_DBG_TIMELINE build writes [wave][visit][8] 100 MHz stamps: 0 top, 1 rows read, 2 body loads issued,
3 first tag check, 4 tags ok, 5 before the publish stores): python tools/visit_stamps.py timeline.bin"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 256, 8)
valid = a[:, :, 0] != 0
t = a[:, :, :6].astype(np.int64) * 10
vt = t[:, :, 0]; nv = valid.sum(1)
d = np.concatenate([np.diff(vt[w, :nv[w]]) for w in range(a.shape[0]) if nv[w] > 1])
nxt = np.concatenate([vt[w, 1:nv[w]] - t[w, :nv[w] - 1, 5] for w in range(a.shape[0]) if nv[w] > 1])
d02 = (t[:, :, 2] - t[:, :, 0])[valid]; d23 = (t[:, :, 3] - t[:, :, 2])[valid]; d34 = (t[:, :, 4] - t[:, :, 3])[valid]; d45 = (t[:, :, 5] - t[:, :, 4])[valid]
print("top -> body loads issued:          mean %7.1f median %7.1f ns" % (d02.mean(), np.median(d02)))
print("body loads issued -> first check:  mean %7.1f median %7.1f p90 %7.1f ns" % (d23.mean(), np.median(d23), np.percentile(d23, 90)))
print("first check -> tags ok (polling):  mean %7.1f median %7.1f p90 %7.1f ns, fraction > 300 ns: %.2f" % (d34.mean(), np.median(d34), np.percentile(d34, 90), (d34 > 300).mean()))
w_ = d34[d34 > 300]
if len(w_):
    print("   ... of the visits that wait:    mean %7.1f median %7.1f p10 %7.1f ns" % (w_.mean(), np.median(w_), np.percentile(w_, 10)))
print("tags ok -> before stores:          mean %7.1f median %7.1f ns" % (d45.mean(), np.median(d45)))
print("before stores -> next top:         mean %7.1f median %7.1f ns" % (nxt.mean(), np.median(nxt)))
print("visit period:                      mean %7.1f median %7.1f ns; waves %d, visits %d, span %.1f us" % (d.mean(), np.median(d), (nv > 0).sum(), valid.sum(), (t[:, :, 5][valid].max() - t[:, :, 0][valid].min()) / 1e3))
