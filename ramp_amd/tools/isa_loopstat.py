"""Instruction mix of the MFMA-densest basic block of every kernel in a hipcc -save-temps .s file (diagnostic)."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end", s, re.S):
    name, body = m.group(1), m.group(2)
    if pat and pat not in name:
        continue
    blocks = re.split(r"\n\.LBB[0-9_]+:", body)
    best = max(blocks, key=lambda b: b.count("v_mfma"))
    ins = [l.strip().split()[0] for l in best.split("\n") if l.strip() and not l.strip().startswith((".", ";"))]
    c = Counter(ins)
    mf = sum(v for k, v in c.items() if k.startswith("v_mfma"))
    valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    print(f"{name[:70]}: mfma {mf} valu {valu} (accvgpr {c['v_accvgpr_read_b32'] + c['v_accvgpr_write_b32']} mov {c['v_mov_b32_e32']}) "
          f"ds_read {sum(v for k, v in c.items() if k.startswith('ds_read'))} waitcnt {c['s_waitcnt']} nop {c['s_nop']} total {len(ins)}")
