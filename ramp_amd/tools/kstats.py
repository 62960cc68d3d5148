"""Print a rocprofv3 --kernel-trace --stats summary (x_kernel_stats.csv) per step: kstats.py file.csv n_steps [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / steps:.1f} ms per step ({steps:g} steps in the trace)")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.1f} ms {float(r['Percentage']):5.1f}% n={int(r['Calls']) / steps:8.1f} "
          f"avg={float(r['AverageNs']) / 1e3:8.1f}us  {r['Name'][:120]}")
