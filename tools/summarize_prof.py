"""Runs on the GPU box: reduce rocprofv3 csv outputs (kernel stats + PMC counter collection) to small summaries."""
import csv, json, sys, collections
from pathlib import Path

raw = Path(sys.argv[1]); out = Path(sys.argv[2]); out.mkdir(parents=True, exist_ok=True)
# 1. kernel stats csv: copy (small)
found = sorted((raw / "stats").glob("*_kernel_stats.csv"))
ks = found[0] if found else raw / "stats" / "missing"
prefix = ks.name.replace("_kernel_stats.csv", "") if found else "r02"
if ks.exists():
    rows = list(csv.reader(open(ks)))
    with open(out / f"{prefix}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:110]
            w.writerow(r)
# 1b. kernel trace of the same pass: per-kernel mean duration of the LAST 20 dispatches = the timed steps of `bench.py --steps 20` (+ its 6 extra
# profiled steps): the figure bench.py's roofline.avg_launch_us has to agree with (the stats csv above averages over the settle phase too)
tr = sorted((raw / "stats").glob("*_kernel_trace.csv"))
if tr:
    dur = collections.defaultdict(list)
    rows = list(csv.DictReader(open(tr[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    steady = {k: {"steady_avg_us": sum(v[-min(20, len(v)):]) / min(20, len(v)), "avg_us": sum(v) / len(v), "dispatches": len(v)} for k, v in dur.items()}
    json.dump(steady, open(out / f"{prefix}_kernel_steady.json", "w"), indent=1)
    print(json.dumps({k: round(v["steady_avg_us"], 1) for k, v in steady.items() if "contact_solve" in k}))
# 2. PMC: per-kernel mean of the counter
summary = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    cand = sorted((raw / ("pmc_fetch" if name == "FETCH_SIZE" else "pmc_write")).glob("*_counter_collection.csv"))
    if not cand:
        continue
    p = cand[0]
    vals = collections.defaultdict(list)
    with open(p) as f:
        rd = csv.DictReader(f)
        for r in rd:
            if r.get("Counter_Name") != name:
                continue
            vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    # steady state = the kernel's last 20 dispatches = the timed steps of `bench.py --steps 20` and its 6 extra steps (the 240 settle steps come before)
    summary[name] = {k: {"mean": sum(v) / len(v), "steady_mean": sum(v[-min(20, len(v)):]) / min(20, len(v)),
                         "dispatches": len(v), "sum": sum(v)} for k, v in vals.items()}
json.dump(summary, open(out / f"{prefix}_pmc_summary.json", "w"), indent=1)
print(json.dumps({n: {k: round(v["steady_mean"], 1) for k, v in d.items() if "contact_solve" in k or "narrow" in k or "pairs_grid" in k} for n, d in summary.items()}))
# HBM traffic of the dominant kernel(s), per launch, steady state.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
# reports half the bytes of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE as reported.
traffic = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (tools/gpu_pmc.sh); per-launch mean over the last "
                     "20 dispatches (timed + profiled steps after the 240 settle steps); bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE on gfx950 reports half "
                     "of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section); WRITE_SIZE used as reported (uncalibrated)"}
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    for k in summary["FETCH_SIZE"]:
        if "contact_solve" in k and k in summary["WRITE_SIZE"]:
            name = k.split("::")[-1].split("<")[0]   # template arguments (k_contact_solve_persist<slot data in LDS, XCD-partitioned>) dropped
            f_, w_ = summary["FETCH_SIZE"][k]["steady_mean"], summary["WRITE_SIZE"][k]["steady_mean"]
            traffic[name + "_bytes_per_launch"] = (2 * f_ + w_) * 1024
            traffic[name + "_raw"] = {"FETCH_SIZE_KB_steady": f_, "WRITE_SIZE_KB_steady": w_, "dispatches": summary["FETCH_SIZE"][k]["dispatches"]}
    # contacts of the profiled state: from the bench line the PMC run itself printed (third argument: its log)
    try:
        line = [l for l in open(sys.argv[3]).read().splitlines() if l.startswith("{")][-1]
        b = json.loads(line)
        contacts, sweeps = b["config"]["contacts"], 20
        traffic["profiled_state"] = {"contacts": contacts, "sweeps": sweeps, "settle_steps": b["config"].get("settle_steps")}
        traffic["source"] = f"{prefix}_pmc_summary.json, {contacts} contacts x {sweeps} sweeps per launch"
        for k in list(traffic):
            if k.endswith("_bytes_per_launch"):
                traffic[k.replace("_bytes_per_launch", "_bytes_per_contact_sweep")] = traffic[k] / (contacts * sweeps)
    except Exception as e:   # noqa: BLE001
        traffic["profiled_state_error"] = str(e)
    json.dump(traffic, open(out / "traffic.json", "w"), indent=1)
