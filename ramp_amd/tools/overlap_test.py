"""Does an HBM-bound row kernel overlap with the MFMA-bound x6 GEMM when launched on a second stream?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["RAMP_GEMM_MODE"] = "bf16x6"
import torch
from ramp_amd import _lib
lib = _lib.load()
M, N, K = 393216, 256, 2048
A = torch.randn(M, K, device="cuda"); W = torch.randn(1, N, K, device="cuda") * 0.05; C = torch.empty(M, N, device="cuda")
X = torch.randn(M, 256, device="cuda"); Y = torch.empty_like(X); g = torch.ones(256, device="cuda"); b = torch.zeros(256, device="cuda")
DY = torch.randn(M, 256, device="cuda"); DX = torch.empty_like(X)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
NG, NL = 10, 60


def gemms(s):
    for _ in range(NG):
        lib.ramp_op_gemm(_lib.ptr(A), _lib.ptr(W), None, None, _lib.ptr(C), M, N, K, 1, 0, 0, 1, s.cuda_stream)


def lns(s):
    for _ in range(NL // 2):
        lib.ramp_op_layernorm(_lib.ptr(X), _lib.ptr(g), _lib.ptr(b), _lib.ptr(Y), M, s.cuda_stream)
        lib.ramp_op_layernorm_bwd(_lib.ptr(DY), _lib.ptr(X), _lib.ptr(g), None, _lib.ptr(DX), M, s.cuda_stream)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3


gemms(sa); lns(sb); torch.cuda.synchronize()
tg = timed(lambda: gemms(sa)); tl = timed(lambda: lns(sb))
tc = timed(lambda: (gemms(sa), lns(sb)))
tc2 = timed(lambda: (lns(sb), gemms(sa)))
print(f"gemm alone {tg:.2f} ms, row kernels alone {tl:.2f} ms, sum {tg + tl:.2f} ms; concurrent {tc:.2f} ms / {tc2:.2f} ms (rows first)")
