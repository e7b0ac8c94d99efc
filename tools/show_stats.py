"""dev helper: print the rocprofv3 kernel-stats csv compactly (per-step averages)."""
import csv, sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_summary/r01_kernel_stats.csv"
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 263.0
rows = list(csv.reader(open(path)))[1:]
tot = sum(float(r[2]) for r in rows)
print(f"total kernel time per step: {tot/1e3/steps:.1f} us")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print(f"{r[0][:58]:58s} calls/step={int(r[1])/steps:7.1f} us/step={float(r[2])/1e3/steps:8.1f} avg_us={float(r[3])/1e3:8.2f} min={float(r[5])/1e3:7.2f} max={float(r[6])/1e3:7.2f}")
