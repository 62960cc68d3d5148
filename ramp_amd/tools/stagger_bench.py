"""Start-phase offsets of co-resident GEMM blocks: plain fp16x3 shapes with no offset and with each variant."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load()


def t(M, N, K, flags, iters=10):
    us = C.c_float()
    _lib.check(lib.ramp_bench_gemm(M, N, K, 1, 1, 3, flags, 3, iters, C.byref(us), None))
    return us.value


for (M, N, K, fl) in [(393216, 256, 256, 3), (393216, 768, 256, 1), (393216, 256, 768, 3), (393216, 1024, 256, 1),
                      (196608, 256, 256, 3), (196608, 768, 256, 1), (49152, 256, 256, 3)]:
    base = min(t(M, N, K, fl) for _ in range(3))
    out = [f"M={M} N={N} K={K}: base {base:.1f} us ({2.0 * M * N * K / base / 1e6:.0f} TF)"]
    for v in range(4):
        for T in (8, 16, 32, 64):
            us = min(t(M, N, K, fl | ((v << 12 | T) << 16)) for _ in range(2))
            out.append(f"v{v}/T{T} {us:.1f}")
    print("  ".join(out), flush=True)
