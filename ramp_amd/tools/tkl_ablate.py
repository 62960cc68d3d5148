"""Where the token-owning linear's time goes: the kernel with parts switched off (timing only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()


def t(M, N, flags, iters=10):
    us = C.c_float()
    _lib.check_tools(lib.ramp_bench_gemm(M, N, 256, 1, 1, 8, flags, 3, iters, C.byref(us), None))
    return us.value


M = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
names = {0: "full", 1: "no weight DMA", 2: "no stores", 3: "no DMA, no stores", 8: "no barrier"}
for N, fl, what in ((768, 1, "LN -> QKV"), (256, 2, "out-proj"), (256, 0, "d(o)")):
    for ab, nm in names.items():
        us = min(t(M, N, fl | (ab << 8)) for _ in range(2))
        print(f"M={M} N={N} {what:10s} {nm:20s} {us:7.1f} us  {2.0 * M * N * 256 / us / 1e6:5.0f} TF", flush=True)
