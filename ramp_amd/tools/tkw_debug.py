"""Where a tkw launch differs from float64: plain convolution / delayed scale / partial tiles / GroupNorm epilogue, one line each;
the error map of the first failing variant by token position and 32-channel block."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ramp_amd import _lib
from test_gpu_ops import _conv5_f64

L, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (6, 256, 256)))
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
lib = _lib.load()
shown = False
for R, prev_f, epi in ((16, 0.0, 0), (16, 0.8, 0), (37, 0.0, 0), (37, 0.8, 0), (16, 0.0, 1), (16, 0.8, 1), (37, 0.8, 1)):
    g = np.random.Generator(np.random.PCG64(3))
    M = L * R
    X = g.standard_normal((M, K)).astype(np.float32)
    W = (g.standard_normal((5, N, K)) / np.sqrt(5 * K)).astype(np.float32)
    bias = g.standard_normal(N).astype(np.float32)
    gam = np.ones(N, np.float32); bet = np.zeros(N, np.float32)
    ref = _conv5_f64(X.astype(np.float64), W.astype(np.float64), L, 1) + bias
    Y = torch.full((M, N), float("nan"), device="cuda"); Cs = torch.full((M, N), float("nan"), device="cuda"); st = torch.zeros((R, 8, 2), device="cuda")
    amax, flag = C.c_float(0), C.c_int32(0)
    dX, dW, db, dg, dbt = d(X), d(W), d(bias), d(gam), d(bet)      # (kept alive: a temporary's memory is recycled by the next allocation)
    _lib.check(lib.ramp_op_tkw(_lib.ptr(dX), None, 0, _lib.ptr(dW), _lib.ptr(db), None, None, None, None, None, None,
                               _lib.ptr(dg) if epi else None, _lib.ptr(dbt) if epi else None, None,
                               M, L, N, K, 1, N, float(np.abs(X).max()) * prev_f, _lib.ptr(Y), None, _lib.ptr(Cs) if epi else None, _lib.ptr(st) if epi else None,
                               C.byref(amax), C.byref(flag), None), "ramp_op_tkw")
    got = (Cs if epi else Y).cpu().numpy()
    err = np.abs(got - ref) / np.abs(ref).max()
    print(f"R={R} prev={prev_f} epi={epi}: max rel err {np.nanmax(err):.2e} nan {np.isnan(got).sum()} flag {flag.value}", flush=True)
    if np.nanmax(err) > 1e-5 and not shown:
        shown = True
        bt = err.reshape(M, N // 32, 32).max(axis=2)
        print("rows = token, cols = 32-channel block; log10 err")
        for t in range(min(M, 96)):
            print(f"{t:3d} " + " ".join(f"{np.log10(max(v, 1e-9)):5.1f}" for v in bt[t]))
        ch = err[: min(M, 96)].max(axis=0).reshape(N // 32, 32)
        print("per channel (block x 32):")
        for b in range(N // 32):
            print(b, " ".join(f"{np.log10(max(v, 1e-9)):5.1f}" for v in ch[b]))
