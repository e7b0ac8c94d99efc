#!/bin/bash
# per-kernel counters at the bench state (separate rocprofv3 --pmc passes, kernel-trace only): where each kernel's time goes
ulimit -c 0; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-at-rest"
rm -rf /tmp/pmc_r3; mkdir -p /tmp/pmc_r3
i=0
for CTR in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmc_r3/p$i -o x -- $B > /tmp/pmc_r3/p$i.log 2>&1 || echo "pass $i ($CTR) failed: $(tail -2 /tmp/pmc_r3/p$i.log | tr '\n' ' ')"
done
python - <<'PY'
import csv, glob, json, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_r3/**/x_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tr = sorted(glob.glob("/tmp/pmc_r3/p1/**/x_kernel_trace.csv", recursive=True))
if tr:
    rows = list(csv.DictReader(open(tr[0]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows: dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k, d in vals.items():
    n = 6
    out[k] = {c: sum(v[-n:]) / len(v[-n:]) for c, v in d.items()}
    out[k]["us_under_pmc"] = sum(dur[k][-n:]) / max(1, len(dur[k][-n:])) if dur.get(k) else None
json.dump(out, open("gpurun_out/r3_pmc_kernels.json", "w"), indent=1)
for k, m in sorted(out.items(), key=lambda kv: -(kv[1].get("us_under_pmc") or 0))[:14]:
    cyc = max(m.get("SQ_WAVE_CYCLES", 0), 1)
    print("%7.1f us %-36s valu %6.1fM issue%% %4.1f wait_any%% %4.1f wait_inst%% %4.1f vmem_rd %6.2fM wr %6.2fM lds %6.2fM  fetch %6.1f MB write %6.1f MB  L2 hit%% %4.1f  atom %.2fM/%.2fM" % (
        m.get("us_under_pmc") or 0, k[:36], m.get("SQ_INSTS_VALU", 0) / 1e6, 100 * m.get("SQ_INSTS_VALU", 0) / cyc, 100 * m.get("SQ_WAIT_ANY", 0) / cyc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / cyc,
        m.get("SQ_INSTS_VMEM_RD", 0) / 1e6, m.get("SQ_INSTS_VMEM_WR", 0) / 1e6, m.get("SQ_INSTS_LDS", 0) / 1e6, 2 * m.get("FETCH_SIZE", 0) / 1024, m.get("WRITE_SIZE", 0) / 1024,
        100 * m.get("TCC_HIT_sum", 0) / max(1, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)), m.get("TCP_TCC_ATOMIC_WITH_RET_REQ_sum", 0) / 1e6, m.get("TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum", 0) / 1e6))
PY
