"""Runs on the GPU box: reduce rocprofv3 csv outputs (kernel stats + PMC counter collection) to small summaries."""
import csv, json, sys, collections
from pathlib import Path

raw = Path(sys.argv[1]); out = Path(sys.argv[2]); out.mkdir(parents=True, exist_ok=True)
# 1. kernel stats csv: copy (small)
ks = raw / "stats" / "r01_kernel_stats.csv"
if ks.exists():
    rows = list(csv.reader(open(ks)))
    with open(out / "r01_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:110]
            w.writerow(r)
# 2. PMC: per-kernel mean of the counter
summary = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    p = raw / ("pmc_fetch" if name == "FETCH_SIZE" else "pmc_write") / "r01_counter_collection.csv"
    if not p.exists():
        continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(p) as f:
        rd = csv.DictReader(f)
        for r in rd:
            if r.get("Counter_Name") != name:
                continue
            k = r["Kernel_Name"].split("(")[0]
            a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
    summary[name] = {k: {"mean": v[0] / v[1], "dispatches": v[1], "sum": v[0]} for k, v in acc.items()}
json.dump(summary, open(out / "r01_pmc_summary.json", "w"), indent=1)
print(json.dumps({n: {k: round(v["mean"], 1) for k, v in d.items() if "contact_solve" in k or "narrow" in k or "pairs_grid" in k} for n, d in summary.items()}))
