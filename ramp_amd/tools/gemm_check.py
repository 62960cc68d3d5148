"""Accuracy check of ramp_op_gemm_mode against float64: gemm_check.py [fp32|bf16x6|bf16x6-lds|fp16x3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from ramp_amd import _lib

lib = _lib.load()
torch.manual_seed(0)
worst = 0.0
for (M, N, K, taps, L) in [(1000, 128, 32, 1, 1), (4096, 256, 256, 1, 1), (777, 160, 64, 1, 1), (48 * 40, 256, 128, 5, 48),
                           (6 * 300, 128, 256, 5, 6), (12 * 100, 512, 64, 3, 12), (300, 2048, 256, 1, 1), (129, 768, 2048, 1, 1)]:
    A = torch.randn(M, K, device="cuda") * (1 + 3 * torch.rand(M, 1, device="cuda"))
    W = torch.randn(taps, N, K, device="cuda") * 0.1
    b = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    sh0, st = (-(taps // 2), 1) if taps > 1 else (0, 0)
    _lib.op_gemm(A, W, b, r, C, M, N, K, taps, sh0, st, L, mode=sys.argv[1] if len(sys.argv) > 1 else "fp16x3")
    A64 = A.double().view(M // L, L, K); ref = r.double() + b.double()
    acc = torch.zeros(M // L, L, N, dtype=torch.float64, device="cuda")
    for t in range(taps):
        sh = sh0 + t * st
        sl = torch.zeros_like(A64)
        if sh >= 0:
            sl[:, :L - sh] = A64[:, sh:]
        else:
            sl[:, -sh:] = A64[:, :L + sh]
        acc += sl @ W[t].double().T
    ref = ref + acc.view(M, N)
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    worst = max(worst, err)
    print(f"M={M} N={N} K={K} taps={taps} L={L}: rel err {err:.2e}", flush=True)
assert worst < 2e-6, worst
print("gemm_check ok", worst)
