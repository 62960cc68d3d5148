"""The narrow k = 5 convolutions on sample-owning waves (tkc.hip, ramp_bench_gemm mode 12) against the kernels they replace (mode 3: the
fp16x3 tile kernel for N = 64, mode 0: the exact-fp32 kernel for N = 32), same box, the bench workload's shapes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()


def t(M, N, K, L, mode, flags):
    best = 1e30
    for _ in range(2):
        us = C.c_float(0)
        _lib.check_tools(lib.ramp_bench_gemm(M, N, K, 5, L, mode, flags, 3, 10, C.byref(us), None), "ramp_bench_gemm")
        best = min(best, us.value)
    return best


for (L, R, N, K) in ((48, 4096, 32, 32), (48, 8192, 32, 32), (24, 8192, 64, 32), (24, 8192, 64, 64), (24, 8192, 32, 64), (24, 8192, 32, 32), (12, 8192, 64, 64)):
    M = L * R
    a = t(M, N, K, L, 12, 1)
    b = t(M, N, K, L, 3 if N >= 64 else 0, 1)
    fl = 2.0 * M * N * K * 5
    by = M * (N + K) * 4.0
    print(f"L={L:2d} rows={R} {K:2d}->{N:2d}: tkc {a:7.1f} us ({fl / a / 1e6:5.1f} TFLOP/s, {by / a / 1e6:4.2f} TB/s)   tile kernel {b:7.1f} us   x{b / a:.2f}", flush=True)
