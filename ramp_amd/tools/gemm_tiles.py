"""A/B of the GEMM tile variants on the transformer shapes (interleaved rounds in one process): gemm_tiles.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
M = 393216


def t(N, K, flags, iters=6, mode=3):
    us = C.c_float()
    _lib.check_tools(lib.ramp_bench_gemm(M, N, K, 1, 1, mode, flags, 2, iters, C.byref(us), None))
    return us.value


shapes = [("qkv", 768, 256, 0), ("out+res", 256, 256, 3), ("ff1 geglu", 2048, 256, 1 | 4), ("ff2+res", 256, 1024, 3),
          ("dhg", 1024, 256, 0), ("ff1 dX amul", 256, 2048, 8), ("d(o)", 256, 256, 0), ("dln1", 256, 768, 0)]
variants = [("wide", 0), ("narrow", 16)]
for name, N, K, fl in shapes:
    best = {}
    for rnd in range(3):
        for vn, vf in variants:
            if (fl & 8) and (vf & 32):
                continue
            us = t(N, K, fl | vf)
            best[vn] = min(best.get(vn, 1e30), us)
        if not (fl & (4 | 8)):
            best["dma stag"] = min(best.get("dma stag", 1e30), t(N, K, fl, mode=4))
            best["dma flat"] = min(best.get("dma flat", 1e30), t(N, K, fl | 16, mode=4))
    fl_tf = 2.0 * M * N * K / 1e12
    print(f"{name:12s} N={N:5d} K={K:5d}: " + "  ".join(f"{vn} {us:7.1f} us ({fl_tf / (us * 1e-6):5.1f} TF)" for vn, us in best.items()), flush=True)
