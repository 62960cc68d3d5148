"""One x6 / fp32 GEMM shape repeated a few times (for rocprofv3 --pmc runs): gemm_pmc.py M N K [resid]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ramp_amd import _lib
lib = _lib.load()
M, N, K = (int(v) for v in sys.argv[1:4])
resid = len(sys.argv) > 4
A = torch.randn(M, K, device="cuda"); W = torch.randn(1, N, K, device="cuda") * 0.05; C = torch.empty(M, N, device="cuda")
R = torch.randn(M, N, device="cuda") if resid else None
for _ in range(5):
    _lib.check(lib.ramp_op_gemm(_lib.ptr(A), _lib.ptr(W), None, _lib.ptr(R), _lib.ptr(C), M, N, K, 1, 0, 0, 1, None))
torch.cuda.synchronize()
