"""Per-KERNEL counter summary of the last fp16x3 evaluation of sample_pmc.py, merged from several `rocprofv3 --pmc` passes
(SQ passes of <= 8 counters each, FETCH_SIZE and WRITE_SIZE in passes of their own -- MI355X_MICROARCH.md, "rocprofv3 PMC slots").

    python3 ramp_amd/tools/pmc_round.py profiles/r05_pmc_kernels.json profiles/r05_pmc_traffic.json <pass dir> [<pass dir> ...]

Per kernel (template arguments kept): launches, time under the collection (the collection holds the shader clock low: only
ratios count), every raw counter summed over the evaluation's launches, and the derived figures
  mfma_pipe_occupancy   SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 1024 SIMDs / 32)          (r02 / r03 formula)
  valu_per_mfma         SQ_INSTS_VALU / SQ_INSTS_MFMA        (SQ_INSTS_VALU includes the MFMAs)
  *_frac_of_wave        SQ_WAIT_ANY (parked at s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall), SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
  lds_bank_conflict_frac  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  hbm_bytes             (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads)
The second output is the per-class traffic file bench.py reads (gemm / attention / norm_rows / other_ramp, `ffx` beside them)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

from pmc_summary import klass


def main():
    out_k, out_t, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = defaultdict(lambda: defaultdict(float)); launches = defaultdict(lambda: defaultdict(set)); dur = defaultdict(dict)
    full = {}
    for d in dirs:
        rows = []
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f, newline="")):
                if "ramp::" in r["Kernel_Name"]:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"]),
                                 int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        if not rows:
            print(f"(no counters under {d})"); continue
        rows.sort()
        ends = sorted({i for i, n, c, v, t in rows if "cfg_mean" in n})
        lo, hi = ends[-2], ends[-1]
        for i, n, c, v, t in rows:
            if not (lo < i <= hi):
                continue
            m = re.search(r"ramp::(?:\(anonymous namespace\)::)?(\w+)(<[^>]*>)?", n)
            if not m:
                continue
            k = m.group(1) + (m.group(2) or "")
            full[k] = n
            agg[k][c] += v; launches[k][c].add(i); dur[k][i] = t
    out = {}
    for k, cs in agg.items():
        e = dict(cs)
        e["launches"] = max(len(s) for s in launches[k].values())
        e["time_us_under_pmc"] = sum(dur[k].values()) / 1e3 / max(1, len(dur[k])) * e["launches"]
        if e.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_pipe_occupancy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * e["SQ_BUSY_CYCLES"] / 32)
        if e.get("SQ_INSTS_MFMA", 0) > 0 and "SQ_INSTS_VALU" in e:
            e["valu_per_mfma"] = e["SQ_INSTS_VALU"] / e["SQ_INSTS_MFMA"]
        if e.get("SQ_WAVE_CYCLES", 0) > 0:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in e:
                    e[c + "_frac_of_wave"] = e[c] / e["SQ_WAVE_CYCLES"]
        if e.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in e:
            e["lds_bank_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            e["hbm_read_bytes"] = 2048.0 * e.get("FETCH_SIZE", 0.0); e["hbm_write_bytes"] = 1024.0 * e.get("WRITE_SIZE", 0.0)
            e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
            e["hbm_bytes_per_launch"] = e["hbm_bytes"] / e["launches"]
        out[k] = e
    out["_note"] = ("the LAST (third, fp16x3) score evaluation of ramp_amd/tools/sample_pmc.py -- B = 4096 trajectories = 8192 network rows, "
                    "the kernels bench.py times -- under separate rocprofv3 --pmc passes (program directly after `--`); the collection "
                    "holds the shader clock low, so only ratios count; hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction)")
    json.dump(out, open(out_k, "w"), indent=1)
    # per-class traffic (the file bench.py reads), with the dominant kernel (ffx) as a class of its own beside the totals
    cls = {}
    for k, e in out.items():
        if not isinstance(e, dict) or "hbm_bytes" not in e:
            continue
        for c in ([klass("ramp::" + k)] + (["ffx"] if k.startswith(("ffx_kernel", "ffx16_kernel", "ffx16h_kernel")) else [])):
            if c is None:
                continue
            q = cls.setdefault(c, {"launches": 0, "read_bytes": 0.0, "write_bytes": 0.0})
            q["launches"] += e["launches"]; q["read_bytes"] += e["hbm_read_bytes"]; q["write_bytes"] += e["hbm_write_bytes"]
    tot = 0.0
    for c, q in cls.items():
        q["hbm_bytes"] = q["read_bytes"] + q["write_bytes"]; q["hbm_bytes_per_launch"] = q["hbm_bytes"] / max(1, q["launches"])
        if c != "ffx":
            tot += q["hbm_bytes"]
    cls["_note"] = out["_note"]; cls["collected"] = "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), ramp_amd/tools/pmc_round.py"
    cls["total_hbm_bytes_per_evaluation"] = tot
    json.dump(cls, open(out_t, "w"), indent=1)
    print(f"{'kernel':58s} {'n':>4s} {'t_us':>9s} {'mfma':>5s} {'v/m':>5s} {'wait':>5s} {'pipe':>5s} {'issue':>5s} {'ldsc':>5s} {'GB':>7s}")
    for k, e in sorted(((k, e) for k, e in out.items() if isinstance(e, dict)), key=lambda kv: -kv[1].get("time_us_under_pmc", 0))[:24]:
        print(f"{k[:58]:58s} {e['launches']:4d} {e['time_us_under_pmc']:9.1f} {e.get('mfma_pipe_occupancy', 0):5.2f} {e.get('valu_per_mfma', 0):5.1f} "
              f"{e.get('SQ_WAIT_ANY_frac_of_wave', 0):5.2f} {e.get('SQ_WAIT_INST_ANY_frac_of_wave', 0):5.2f} {e.get('SQ_ACTIVE_INST_ANY_frac_of_wave', 0):5.2f} "
              f"{e.get('lds_bank_conflict_frac', 0):5.2f} {e.get('hbm_bytes', 0) / 1e9:7.2f}")
    print(f"total HBM bytes per evaluation: {tot / 1e9:.1f} GB")


if __name__ == "__main__":
    main()
