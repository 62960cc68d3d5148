"""Where a kernel waits for ALL (or nearly all) of its vector-memory operations and where it touches scratch: per basic
block, the s_waitcnt vmcnt(N <= limit) and scratch operations with the MFMA count before them.  A vmcnt(0) inside a loop
that also carries inline-asm LDS-DMA (or any prefetch) is a full memory round trip; a scratch reload is a vmcnt(0) too.
isa_waits.py file.s kernel-name-substring [limit=2]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 2
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l.split(":")[0]]
for start in starts:
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    print(lines[start].split(":")[0])
    blk, n = "entry", 0
    per = {}
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blk, n = m.group(1), 0
        e = per.setdefault(blk, {"mfma": 0, "waits": Counter(), "scratch": 0, "loop": False, "lines": 0})
        e["lines"] += 1
        if "v_mfma" in t:
            n += 1
            e["mfma"] = n
        w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if w and int(w.group(1)) <= limit:
            e["waits"][int(w.group(1))] += 1
        if "scratch_" in t:
            e["scratch"] += 1
        if re.search(r"s_cbranch\w*\s+" + re.escape(blk) + r"\b", t):
            e["loop"] = True
    for b, e in per.items():
        if e["waits"] or e["scratch"]:
            print(f"  {b:12s} lines {e['lines']:5d} mfma {e['mfma']:4d} {'LOOP' if e['loop'] else '    '} small waits {dict(e['waits'])} scratch {e['scratch']}")
