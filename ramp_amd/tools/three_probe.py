import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from ramp_amd import _lib
lib = _lib.load_tools()
def t(M, N, K, taps, L, mode, flags, iters=20):
    us = C.c_float(); _lib.check_tools(lib.ramp_bench_gemm(M, N, K, taps, L, mode, flags, 3, iters, C.byref(us), None)); return us.value
for (M, N, K) in ((49152, 256, 256), (49152, 768, 256), (98304, 256, 256), (49152, 256, 768), (49152, 1024, 256), (24576, 256, 256), (24576, 768, 256)):
    for fl, name in ((1, "bias"), (3, "bias+resid"), (0, "none")):
        a = min(t(M, N, K, 1, 1, 3, fl) for _ in range(3)); b = min(t(M, N, K, 1, 1, 3, fl | 32) for _ in range(3))
        print(f"M={M} N={N} K={K} {name}: auto {a:.1f} us, 3 blocks/CU {b:.1f} us  x{a/b:.2f}", flush=True)
