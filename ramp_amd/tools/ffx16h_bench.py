"""ffx16.hip's tilings on one box: full tiles only (flags 1 << 17), half tiles only (2 << 17), launch_ffx16's own policy (0) -- ramp_bench_gemm modes 6 / 7
with flags bit 16 (the 16-wide pair).  usage: python ramp_amd/tools/ffx16h_bench.py [rounds]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
FL = 2.0 * (2048 * 256 + 256 * 1024)
for rnd in range(rounds):
    for M in (49152, 24576, 393216, 16384 + 32768):
        for mode, name in ((6, "forward"), (7, "backward")):
            row = []
            for hm, tag in ((1, "full"), (2, "half"), (0, "auto")):
                us = C.c_float()
                _lib.check_tools(lib.ramp_bench_gemm(M, 2048, 256, 1, 1, mode, (1 << 16) | (hm << 17), 30, 120, C.byref(us), None), "bench")
                row.append(f"{tag} {us.value:8.1f} us {FL * M / us.value / 1e6:6.1f} TF")
            print(f"round {rnd} M={M:6d} {name:8s}: " + "   ".join(row), flush=True)
