"""Attention backward on sample-owning waves (atk.hip atb_kernel, ramp_bench_gemm mode 13) against attn2_bwd_kernel (mode 14), same box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
for L, R in ((48, 4096), (48, 8192), (24, 8192), (12, 8192), (6, 8192)):
    M = L * R
    row = []
    for mode in (13, 14):
        best = 1e30
        for _ in range(2):
            us = C.c_float(0)
            _lib.check_tools(lib.ramp_bench_gemm(M, 256, 256, 1, L, mode, 0, 3, 10, C.byref(us), None), "ramp_bench_gemm")
            best = min(best, us.value)
        row.append(best)
    by = M * 7168.0
    print(f"L={L:3d} rows={R:5d} tokens={M:7d}: atb {row[0]:8.1f} us ({by / row[0] / 1e6:4.2f} TB/s)   attn2_bwd {row[1]:8.1f} us ({by / row[1] / 1e6:4.2f} TB/s)   x{row[1] / row[0]:.2f}", flush=True)
