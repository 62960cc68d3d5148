"""Print per-counter averages of the GEMM dispatches in a rocprofv3 --pmc CSV directory."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_x6" in r["Kernel_Name"] or "gemm_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 2:]
    print(f"{k:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
