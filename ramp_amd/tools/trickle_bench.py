"""Output stores issued in a burst at the end of each tile (the product kernel) vs two per K slab inside the next tile's loop
(diagnostic twin, garbage data): does a smooth store stream let the loads, the MFMAs and the stores overlap?"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()


def t(M, N, K, flags, iters=10):
    us = C.c_float()
    _lib.check_tools(lib.ramp_bench_gemm(M, N, K, 1, 1, 3, flags, 3, iters, C.byref(us), None))
    return us.value


M = 393216
TR = 1 << 30
VARIANTS = [("full", 0), ("no stores", 32 << 8), ("stores inside the loop", (32 << 8) | TR), ("no MFMA", 128 << 8),
            ("no MFMA, stores inside the loop", (160 << 8) | TR), ("MFMA + LDS reads only", 15 << 8)]
for (N, K, fl) in [(256, 256, 3), (256, 256, 1), (768, 256, 1), (1024, 256, 1), (256, 768, 3), (256, 1024, 3)]:
    out = [f"M={M} N={N} K={K} resid={bool(fl & 2)}:"]
    for name, v in VARIANTS:
        us = min(t(M, N, K, fl | v) for _ in range(3))
        out.append(f"{name} {us:.0f} us ({2.0 * M * N * K / us / 1e6:.0f} TF)")
    print(" | ".join(out), flush=True)
