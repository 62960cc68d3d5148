"""Two against three resident blocks per CU for the bias-only and the A-multiplier fp16x3 kernels (64 x 256 tile): the table
behind launch_x6's rule (gemm.hip) for when the third block is used."""
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


for M in (393216, 196608, 98304, 49152):
    for (N, K, fl, name) in [(768, 256, 1, "bias-only"), (1024, 256, 1, "bias-only"), (256, 256, 1, "bias-only"), (256, 768, 1, "bias-only"),
                             (256, 1024, 1, "bias-only"), (256, 2048, 8, "A-multiplier")]:
        a = min(t(M, N, K, fl | 16 * 0) for _ in range(3))       # the launch rule's own choice
        b2 = min(t(M, N, K, fl | (1 << 8) * 0) for _ in range(1))
        b = min(t(M, N, K, fl | 32) for _ in range(3))           # forced third block
        print(f"{name} M={M} N={N} K={K}: rule {a:.1f} us ({2.0 * M * N * K / a / 1e6:.0f} TF)  forced 3 blocks/CU {b:.1f} us "
              f"({2.0 * M * N * K / b / 1e6:.0f} TF)", flush=True)
