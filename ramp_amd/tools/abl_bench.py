"""Attention backward + d(ln1) + LayerNorm-1 backward as one launch (atl.hip abl_kernel, ramp_bench_gemm mode 15) against the pair it
replaces (atb_kernel + tklb_kernel, mode 16), same box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
cases = ((48, 4096), (48, 8192), (24, 8192), (12, 8192), (6, 8192))
if len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])] if sys.argv[1].isdigit() else cases
for L, R in cases:
    M = L * R
    row = []
    for mode in (15, 16):
        best = 1e30
        for _ in range(2):
            us = C.c_float(0)
            _lib.check_tools(lib.ramp_bench_gemm(M, 256, 256, 1, L, mode, 0, 3, 10, C.byref(us), None), "ramp_bench_gemm")
            best = min(best, us.value)
        row.append(best)
    print(f"L={L:3d} rows={R:5d} tokens={M:7d}: abl {row[0]:8.1f} us   atb + tklb {row[1]:8.1f} us   x{row[1] / row[0]:.2f}", flush=True)
if "--stamps" in sys.argv:
    us = C.c_float(0)
    _lib.check_tools(lib.ramp_bench_gemm(48 * 8192, 256, 256, 1, 48, 15, 256, 2, 5, C.byref(us), None), "ramp_bench_gemm")
    print(f"stamped twin: {us.value:.1f} us", flush=True)
