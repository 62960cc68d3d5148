"""tkw.hip (wide k = 5 convolution on sample-owning blocks, GroupNorm fused) against the tile kernel it replaces, same box:
ramp_bench_gemm mode 17 (tkw; flags 1 GroupNorm + Mish epilogue, 2 GroupNorm-backward operand, 4 input gradient, 8 residual) vs mode 3
(fp16x3 tile kernel, 5 taps; its GroupNorm launch of ~15-35 us comes on top).  tkw_bench.py [rows]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192


def t(M, N, K, L, mode, flags, taps=5):
    us = C.c_float(0)
    _lib.check_tools(lib.ramp_bench_gemm(M, N, K, taps, L, mode, flags, 5, 30, C.byref(us), None), "ramp_bench_gemm")
    return us.value


print(f"{'shape':28s} {'tile conv':>10s} {'tkw plain':>10s} {'tkw+GN fwd':>11s} {'tkw GNbwd':>10s}   TFLOP/s (tile / fwd / bwd)")
for L, K, N in ((6, 256, 256), (6, 128, 256), (6, 512, 128), (6, 128, 128), (12, 128, 128), (12, 64, 128), (6, 256, 128), (6, 128, 512), (8, 256, 256), (16, 128, 128)):
    M = R * L
    fl = 2.0 * M * N * K * 5
    a = t(M, N, K, L, 3, 1)
    b = t(M, N, K, L, 17, 16)
    f = t(M, N, K, L, 17, 1 | 8) if N != 512 else float("nan")
    g = t(M, N, K, L, 17, 2 | 4 | 8) if K <= 256 else float("nan")
    print(f"L={L:2d} {K:3d}->{N:3d} M={M:7d}     {a:10.1f} {b:10.1f} {f:11.1f} {g:10.1f}   {fl / a / 1e6:6.1f} / {fl / f / 1e6:6.1f} / {fl / g / 1e6:6.1f}", flush=True)

# the narrow levels: tkc.hip plain (bias + residual) / with the GroupNorm + Mish epilogue / input gradient with the GroupNorm-backward operand
print(f"\n{'shape':28s} {'tkc plain':>10s} {'tkc+GN fwd':>11s} {'tkc bwd':>10s} {'tkc GNbwd':>10s}")
for L, K, N in ((48, 32, 32), (24, 32, 64), (24, 64, 64)):
    M = R * L
    a = t(M, N, K, L, 12, 1 | 2)
    f = t(M, N, K, L, 12, 2 | 8)
    b = t(M, N, K, L, 12, 4 | 2)
    g = t(M, N, K, L, 12, 4 | 16 | 2) if N >= K else float("nan")
    print(f"L={L:2d} {K:3d}->{N:3d} M={M:7d}     {a:10.1f} {f:11.1f} {b:10.1f} {g:10.1f}", flush=True)
