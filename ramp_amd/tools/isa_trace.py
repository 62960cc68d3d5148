"""Instruction-class trace of one kernel in a hipcc -save-temps .s file: where the MFMAs sit between the vector, LDS,
memory and scalar instructions (M mfma, v valu, T transcendental, a accvgpr move, d LDS, L LDS-DMA, G global, S scratch,
s salu, w lgkm wait, W vm wait, B barrier, n nop, | label).  usage: isa_trace.py file.s kernel_substring [first_line last_line]"""
import sys
import textwrap

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(body))
out, counts = [], {}
for n, l in enumerate(body):
    if not (lo <= n < hi):
        continue
    l = l.strip()
    if l.startswith(".LBB"):
        out.append("|")
        continue
    if not l or l[0] in ";.":
        continue
    op = l.split()[0]
    if op.startswith("v_mfma"): c = "M"
    elif op.startswith("s_waitcnt"): c = "w" if "vmcnt" not in l else "W"
    elif op.startswith("s_barrier"): c = "B"
    elif op.startswith("ds_"): c = "d"
    elif "load_lds" in op: c = "L"
    elif op.startswith("scratch_"): c = "S"
    elif op.startswith(("global_", "buffer_", "flat_")): c = "G"
    elif op.startswith("v_accvgpr"): c = "a"
    elif op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log")): c = "T"
    elif op.startswith("v_"): c = "v"
    elif op.startswith("s_nop"): c = "n"
    elif op.startswith("s_"): c = "s"
    else: c = "?"
    out.append(c)
    counts[c] = counts.get(c, 0) + 1
print(f"{key}: body lines {len(body)}, classes {dict(sorted(counts.items()))}")
print("\n".join(textwrap.wrap("".join(out), 160)))
