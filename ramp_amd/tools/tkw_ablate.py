"""Where a tkw tile's time goes: the kernel with its MFMA loop (1), its operand loads (2) or its epilogue stores (4) switched off
(ramp_bench_gemm mode 17, flags >> 8; wrong results, timing only).  tkw_ablate.py [rows]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192


def t(M, N, K, L, flags):
    us = C.c_float(0)
    _lib.check_tools(lib.ramp_bench_gemm(M, N, K, 5, L, 17, flags, 5, 30, C.byref(us), None), "ramp_bench_gemm")
    return us.value


for name, base in (("plain", 16), ("GN fwd", 1 | 8), ("GN bwd", 2 | 4 | 8)):
    print(name)
    for L, K, N in ((6, 256, 256), (12, 128, 128), (6, 128, 256)):
        M = R * L
        row = [t(M, N, K, L, base | (ab << 8)) for ab in (0, 1, 2, 4, 1 | 2 | 4, 2 | 4)]
        print(f"  L={L:2d} {K:3d}->{N:3d}: full {row[0]:6.1f} | no MFMA {row[1]:6.1f} | no loads {row[2]:6.1f} | no stores {row[3]:6.1f} | none of the three {row[4]:6.1f} | MFMA only {row[5]:6.1f}", flush=True)
