"""The fused FF1 -> GEGLU -> FF2 kernel against its two stand-alone launches, and with one role switched off."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()


def t(M, N, K, mode, flags, iters=5):
    us = C.c_float()
    _lib.check_tools(lib.ramp_bench_gemm(M, N, K, 1, 1, mode, flags, 2, iters, C.byref(us), None))
    return us.value


for M in (393216, 196608, 98304, 49152):
    ff1 = min(t(M, 2048, 256, 3, 1 | 4) for _ in range(2))
    ff2 = min(t(M, 256, 1024, 3, 3) for _ in range(2))
    fused = min(t(M, 2048, 256, 5, 0) for _ in range(2))
    g1only = min(t(M, 2048, 256, 5, 1 << 8) for _ in range(2))
    loop = min(t(M, 2048, 256, 5, 3 << 8) for _ in range(2))
    noff2 = min(t(M, 2048, 256, 5, 4 << 8) for _ in range(2))
    nogeglu = min(t(M, 2048, 256, 5, 8 << 8) for _ in range(2))
    noprio = min(t(M, 2048, 256, 5, 16 << 8) for _ in range(2))
    fl = 2.0 * M * (256 * 2048 + 1024 * 256) / 1e12
    print(f"M={M}: FF1 {ff1:.0f} us + FF2 {ff2:.0f} us = {ff1 + ff2:.0f} us ({fl / ((ff1 + ff2) * 1e-6):.0f} TF); fused {fused:.0f} us "
          f"({fl / (fused * 1e-6):.0f} TF); FF2 role idle {g1only:.0f} us; G1 main loop only {loop:.0f} us; no FF2 phase {noff2:.0f}; no GEGLU phase {nogeglu:.0f}; "
          f"without priority {noprio:.0f}", flush=True)
