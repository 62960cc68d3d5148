"""Per basic block of one kernel in a hipcc -save-temps .s file: MFMAs, scratch (spill) operations, memory operations.
isa_blocks.py file.s kernel-name-substring"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
seg, cur = [], ["entry", 0, 0, 0, 0]
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        seg.append(cur)
        cur = [m.group(1), 0, 0, 0, 0]
    cur[4] += 1
    if "v_mfma" in l:
        cur[1] += 1
    if "scratch_" in l:
        cur[2] += 1
    if re.search(r"\b(global|ds|buffer)_", l):
        cur[3] += 1
seg.append(cur)
print(lines[start].split(":")[0])
print(f"{'block':14s} {'lines':>6s} {'mfma':>5s} {'scratch':>8s} {'mem':>5s}")
for x in seg:
    print(f"{x[0]:14s} {x[4]:6d} {x[1]:5d} {x[2]:8d} {x[3]:5d}")
