"""Data movement alone: the plain fp16x3 kernel's diagnostic twin with the MFMAs removed (operands still fetched, split,
staged, and the epilogue run on zero accumulators), and with parts of that data movement removed as well."""
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
VARIANTS = [("full", 0), ("MFMA + LDS reads only", 15), ("no epilogue", 8), ("no MFMA", 128), ("no MFMA, no epilogue", 136),
            ("no MFMA, no A staging", 129), ("no MFMA, no W loads", 130), ("no MFMA, epilogue only", 131),
            ("no MFMA, no stores", 160), ("no MFMA, no transpose", 192), ("no MFMA, shallow loader", 144)]
for (N, K, fl) in [(256, 256, 3), (768, 256, 1), (1024, 256, 1), (256, 1024, 3)]:
    by = 4.0 * M * (K + N + (N if fl & 2 else 0))
    out = [f"M={M} N={N} K={K} resid={bool(fl & 2)} ({by / 1e9:.2f} GB):"]
    for name, v in VARIANTS:
        us = min(t(M, N, K, fl | (v << 8)) for _ in range(3))
        out.append(f"{name} {us:.0f}")
    print(" | ".join(out), flush=True)
