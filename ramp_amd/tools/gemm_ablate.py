"""Where a GEMM tile's time goes: the plain fp16x3 kernel with parts of its loop switched off (diagnostic twin kernel,
ramp_bench_gemm flags bits 8..11: 1 no A staging, 2 no weight-fragment loads, 4 no slab barrier, 8 no epilogue)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
M = 393216


def t(N, K, flags, iters=6):
    us = C.c_float()
    _lib.check_tools(lib.ramp_bench_gemm(M, N, K, 1, 1, 3, flags, 2, iters, C.byref(us), None))
    return us.value


names = {0: "full", 1: "-A", 2: "-B", 8: "-epi", 32: "-stores", 64: "-transpose", 15: "mfma only"}
for (N, K) in [(256, 256), (768, 256), (1024, 256), (256, 1024)]:
    for tile, tf in (("wide", 0),):
        best = {}
        for rnd in range(2):
            for ab in names:
                us = t(N, K, tf | (ab << 8))
                best[ab] = min(best.get(ab, 1e30), us)
        fl = 2.0 * M * N * K / 1e12
        print(f"N={N:5d} K={K:5d} {tile:6s}: " + "  ".join(f"{names[ab]} {us:7.1f}us/{fl / (us * 1e-6):4.0f}TF" for ab, us in best.items()), flush=True)
