"""Where the token-owning fused feed-forward's time goes: the kernel with parts switched off (timing only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()


def t(M, mode, flags=0, iters=4):
    us = C.c_float()
    _lib.check_tools(lib.ramp_bench_gemm(M, 2048, 256, 1, 1, mode, flags, 2, iters, C.byref(us), None))
    return us.value


M = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
names = {0: "full", 1: "no weight DMA", 2: "no stash traffic", 64: "stamped", 66: "stamped, no stash"}
fl = 2.0 * M * (256 * 2048 + 1024 * 256) / 1e12
for mode, nm in ((6, "fwd"), (7, "bwd")):
    for ab, what in names.items():
        us = min(t(M, mode, ab << 8) for _ in range(2))
        print(f"M={M} {nm} {what:32s} {us:7.0f} us  {fl / us * 1e6:5.0f} TF", flush=True)
