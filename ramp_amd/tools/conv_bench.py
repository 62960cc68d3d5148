import ctypes as C, os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/ramp_amd") else os.getcwd())
from ramp_amd import _lib
lib = _lib.load_tools()
def t(M, N, K, taps, L, mode, flags, iters=10):
    us = C.c_float(); _lib.check_tools(lib.ramp_bench_gemm(M, N, K, taps, L, mode, flags, 3, iters, C.byref(us), None)); return us.value
for (M, N, K, L) in ((196608, 64, 64, 24), (98304, 128, 128, 12), (49152, 256, 256, 6), (98304, 128, 256, 12), (49152, 256, 512, 6), (196608, 64, 128, 24)):
    a = min(t(M, N, K, 5, L, 3, 1) for _ in range(2)); b = min(t(M, N, K, 5, L, 3, 1 | 32) for _ in range(2))
    fl = 2.0 * M * N * K * 5
    print(f"conv5 M={M} N={N} K={K} L={L}: 2 blocks/CU {a:.1f} us ({fl/a/1e6:.0f} TF), 3 blocks/CU {b:.1f} us ({fl/b/1e6:.0f} TF)", flush=True)
