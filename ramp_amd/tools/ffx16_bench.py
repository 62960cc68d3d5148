"""ffx.hip (32x32x16 MFMAs) against ffx16.hip (16x16x32) on the same box: ramp_bench_gemm modes 6 / 7, flags bit 16 = the 16-wide pair,
flags 64 << 8 = the stamped twin (whole-kernel in-kernel clock on stderr).  usage: python ramp_amd/tools/ffx16_bench.py [rounds]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
FL = 2.0 * (2048 * 256 + 256 * 1024)
for rnd in range(rounds):
    for M in (393216, 196608, 49152):
        for mode, name in ((6, "forward"), (7, "backward")):
            row = []
            for flags, tag in ((0, "32x32x16"), (1 << 16, "16x16x32")):
                us = C.c_float()
                _lib.check_tools(lib.ramp_bench_gemm(M, 2048, 256, 1, 1, mode, flags, 30, 120, C.byref(us), None), "bench")
                row.append(f"{tag} {us.value:8.1f} us {FL * M / us.value / 1e6:6.1f} TF")
            print(f"round {rnd} M={M:6d} {name:8s}: " + "   ".join(row), flush=True)
print("stamped twins (clock):", flush=True)
for mode in (6, 7):
    for flags in (64 << 8, (64 << 8) | (1 << 16)):
        us = C.c_float()
        _lib.check_tools(lib.ramp_bench_gemm(393216, 2048, 256, 1, 1, mode, flags, 100, 200, C.byref(us), None), "bench")
        print(f"mode {mode} flags {flags:#x}: {us.value:.1f} us", flush=True)
