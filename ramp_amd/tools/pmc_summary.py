"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs of score_pmc.py into per-kernel-class HBM traffic.

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane coalesced reads
(MI355X_MICROARCH.md, HBM section), so reads are doubled: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Only the LAST score pass is counted (the dispatches between the last two cfg_mean launches).
"""
import csv
import glob
import json
import sys


def klass(name):
    if "gemm_x6" in name or "gemm_kernel" in name or "ff_fwd_kernel" in name or "ffx_kernel" in name or "ffx16_kernel" in name or "ffx16h_kernel" in name or "tkl_kernel" in name or "tkl16_kernel" in name or "tklb_kernel" in name \
            or "ato_kernel" in name or "abl_kernel" in name or "tkc_kernel" in name:      # (ato / abl: attention fused with a GEMM, counted with it)
        return "gemm"
    if "attn2" in name or "atb_kernel" in name:
        return "attention"
    if "gn_" in name or "ln_" in name or "expand_rows" in name or "combine_rows" in name:
        return "norm_rows"
    if "ramp::" in name:
        return "other_ramp"
    return None


def load(d, counter):
    """Dispatches of the LAST score pass: those between the last two cfg_mean launches (one ends each pass)."""
    rows = []
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and "ramp::" in r["Kernel_Name"]:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "cfg_mean" in r[1]]
    assert len(ends) >= 2, "need two score passes"
    last = rows[ends[-2] + 1:ends[-1] + 1]
    return [(i, klass(n), v, 0) for i, n, v in last if klass(n)]


def main():
    fetch, write, out = sys.argv[1:4]
    res = {}
    for d, counter, scale in ((fetch, "FETCH_SIZE", 2.0), (write, "WRITE_SIZE", 1.0)):
        for _, k, v, _ in load(d, counter):
            e = res.setdefault(k, {"launches": {}, "read_bytes": 0.0, "write_bytes": 0.0})
            e["launches"][counter] = e["launches"].get(counter, 0) + 1
            e["read_bytes" if counter == "FETCH_SIZE" else "write_bytes"] += v * 1024.0 * scale
    for k, e in list(res.items()):
        n = max(e["launches"].values())
        assert len(set(e["launches"].values())) == 1, e["launches"]
        e["launches"] = n
        e["hbm_bytes"] = e["read_bytes"] + e["write_bytes"]
        e["hbm_bytes_per_launch"] = e["hbm_bytes"] / n
    res["_note"] = ("one fp16x3 score evaluation (forward + dX backward; the kernels bench.py times) of the headline workload, "
                    "B=4096 trajectories = 8192 network rows; read bytes = 2 x FETCH_SIZE KiB (gfx950 correction), write bytes "
                    "= WRITE_SIZE KiB; counters collected in separate --pmc passes")
    res["collected"] = sys.argv[4] if len(sys.argv) > 4 else "separate rocprofv3 --pmc passes"
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    tot = sum(v["hbm_bytes"] for v in res.values() if isinstance(v, dict))
    res["total_hbm_bytes_per_evaluation"] = tot
    print(json.dumps({k: (v if not isinstance(v, dict) else {kk: round(vv) for kk, vv in v.items()}) for k, v in res.items()}, indent=1))
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
