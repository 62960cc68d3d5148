"""Micro-benchmark of the MFMA GEMM through the C ABI (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ramp_amd import _lib

lib = _lib.load()


def bench(M, N, K, taps=1, L=1, bias=False, resid=False, iters=10):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(taps, N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda") if bias else None
    r = torch.randn(M, N, device="cuda") if resid else None
    C = torch.empty(M, N, device="cuda")
    s = _lib.current_stream()
    sh0, st = (-(taps // 2), 1) if taps > 1 else (0, 0)
    for _ in range(3):
        _lib.check(lib.ramp_op_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(r), _lib.ptr(C), M, N, K, taps, sh0, st, L, s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.ramp_op_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(r), _lib.ptr(C), M, N, K, taps, sh0, st, L, s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    tf = 2.0 * M * N * K * taps / (us * 1e-6) / 1e12
    print(f"M={M:7d} N={N:5d} K={K:5d} taps={taps} bias={int(bias)} resid={int(resid)}: {us:9.1f} us  {tf:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    M = 393216
    bench(4096, 4096, 4096)
    bench(8192, 8192, 8192, iters=3)
    for (N, K) in [(256, 256), (256, 2048), (2048, 256), (1024, 256), (256, 1024), (768, 256), (256, 768)]:
        for (b, r) in [(False, False), (True, True)]:
            bench(M, N, K, bias=b, resid=r)
    bench(49152, 256, 256, taps=5, L=6, bias=True)
    bench(M, 32, 32, taps=5, L=48, bias=True)
    bench(M // 2, 64, 64, taps=5, L=24, bias=True)
