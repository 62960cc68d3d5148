"""Print per-counter averages of the dispatches whose kernel name contains argv[2] (default: the GEMM kernels) in a
rocprofv3 --pmc CSV directory, the dispatch duration and the effective shader clock (GRBM_GUI_ACTIVE / 8 / duration)."""
import csv, glob, sys
from collections import defaultdict
keys = sys.argv[2].split(",") if len(sys.argv) > 2 else ["gemm_x6", "gemm_kernel"]
acc = defaultdict(list)
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in keys):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {}
for k, v in sorted(acc.items()):
    v = v[len(v) // 2:]
    res[k] = sum(v) / len(v)
    print(f"{k:32s} {res[k]:16.1f}  (n={len(v)})")
if "GRBM_GUI_ACTIVE" in res:
    print(f"effective clock {res['GRBM_GUI_ACTIVE'] / 8 / res['_dur_ns']:.3f} GHz")
if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "GRBM_GUI_ACTIVE" in res:
    print(f"MFMA busy / (GUI_ACTIVE / 8 * 1024 SIMDs) = {res['SQ_VALU_MFMA_BUSY_CYCLES'] / (res['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
