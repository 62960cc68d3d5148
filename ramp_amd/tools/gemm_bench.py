"""Micro-benchmark of the GEMM kernels through ramp_bench_gemm (run on the GPU box): gemm_bench.py [mode ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()


def bench(M, N, K, mode, taps=1, L=1, flags=0, iters=10, warmup=3):
    us = C.c_float()
    _lib.check_tools(lib.ramp_bench_gemm(M, N, K, taps, L, _lib.GEMM_MODES[mode], flags, warmup, iters, C.byref(us), None))
    tf = 2.0 * M * N * K * taps / (us.value * 1e-6) / 1e12
    print(f"{mode:8s} M={M:7d} N={N:5d} K={K:5d} taps={taps} flags={flags}: {us.value:9.1f} us  {tf:6.1f} TFLOP/s", flush=True)
    return us.value


if __name__ == "__main__":
    modes = sys.argv[1:] or ["fp16x3", "bf16x6", "fp32"]
    M = 393216
    for mode in modes:
        bench(8192, 8192, 8192, mode, iters=3)
        for (N, K) in [(256, 256), (256, 2048), (2048, 256), (1024, 256), (256, 1024), (768, 256), (256, 768)]:
            for fl in (0, 3):
                bench(M, N, K, mode, flags=fl)
        bench(M, 2048, 256, mode, flags=1 | 4)
        bench(M, 256, 2048, mode, flags=8)
        bench(49152, 256, 256, mode, taps=5, L=6, flags=1)
        bench(M // 2, 64, 64, mode, taps=5, L=24, flags=1)
