"""Per-kernel SQ counter totals of the LAST score pass of score_pmc.py (rocprofv3 --pmc <SQ counters> CSV directory)."""
import csv, glob, json, re, sys
from collections import defaultdict

rows = []
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if "ramp::" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"]),
                         int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
ends = sorted({d for d, n, c, v, t in rows if "cfg_mean" in n})
lo, hi = ends[-2], ends[-1]
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set); dur = defaultdict(dict)
for d, n, c, v, t in rows:
    if not (lo < d <= hi):
        continue
    m = re.search(r"ramp::(\w+)(<[^>]*>)?", n)
    k = m.group(1) + (m.group(2) or "")
    agg[k][c] += v; cnt[k].add(d); dur[k][d] = t
out = {}
for k, cs in agg.items():
    e = {c: v for c, v in cs.items()}
    e["launches"] = len(cnt[k]); e["time_us_under_pmc"] = sum(dur[k].values()) / 1e3
    if "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"] > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        e["mfma_pipe_occupancy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * e["SQ_BUSY_CYCLES"] / 32)
    if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"] > 0:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in e:
                e[c + "_frac_of_wave"] = e[c] / e["SQ_WAVE_CYCLES"]
    out[k] = e
out["_note"] = ("one score evaluation (forward + dX backward) of the headline workload under rocprofv3 --pmc; the collection "
                "runs hold the shader clock near 1.4 GHz, so times are longer than in the timed runs; the evaluation is the last "
                "(third) ramp_score call of score_pmc.py, i.e. the fp16x3 kernels bench.py times (NP = 2)")
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("time_us_under_pmc", 0) if isinstance(kv[1], dict) else 0)[:10]:
    if isinstance(e, dict):
        print(f"{k:50s} n={e['launches']:4d} t={e['time_us_under_pmc']:9.1f}us mfma={e.get('mfma_pipe_occupancy', 0):.2f} "
              f"wait={e.get('SQ_WAIT_ANY_frac_of_wave', 0):.2f} pipewait={e.get('SQ_WAIT_INST_ANY_frac_of_wave', 0):.2f} issue={e.get('SQ_ACTIVE_INST_ANY_frac_of_wave', 0):.2f}")
