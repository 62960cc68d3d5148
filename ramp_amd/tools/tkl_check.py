"""The token-owning K = 256 linear (tkl.hip): parity against float64 and timing against the tile kernels it replaces."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
dev = "cuda:0"


def check(M, N, ln, epi, seed=0, L=6):
    """epi: 0 plain, 1 bias + residual, 3 bias + residual + row-variant bias"""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(dev)
    X, W = r(M, 256, sc=1.3) + 0.2, r(N, 256, sc=1 / 16)
    bias = r(N, sc=0.3) if epi else None
    resid = r(M, N) if epi & 1 else None
    n_var = 3
    rowbias = r(n_var, N, sc=0.5) if epi & 2 else None
    rows = (M + L - 1) // L
    rowvar = (torch.arange(rows, device=dev) % n_var).to(torch.int32) if epi & 2 else None
    g, b = (1 + r(256, sc=0.1), r(256, sc=0.1)) if ln else (None, None)
    xd = X.double()
    if ln:
        xd = torch.nn.functional.layer_norm(xd, (256,), g.double(), b.double(), 1e-5)
    ref = xd @ W.double().T
    if bias is not None:
        ref = ref + bias.double()
    if resid is not None:
        ref = ref + resid.double()
    if rowbias is not None:
        ref = ref + rowbias.double()[rowvar.long()[torch.arange(M, device=dev) // L]]
    xmax = xd.abs().max().item()
    Y = torch.full((M, N), float("nan"), device=dev)
    out, flag = C.c_float(0), C.c_int32(0)
    p = lambda t_: _lib.ptr(t_) if t_ is not None else None
    worst = 0.0
    for prev in (0.0, xmax):            # unscaled operand, then scaled from the maximum
        Y.fill_(float("nan"))
        _lib.check_tools(lib.ramp_op_tkl(p(X), p(W), p(bias), p(resid), p(rowbias), p(rowvar), n_var if epi & 2 else 0, L, p(g), p(b), M, N,
                                   prev, p(Y), C.byref(out), C.byref(flag), None), "ramp_op_tkl")
        e = (Y.double() - ref).abs().max().item() / ref.abs().max().item()
        print(f"M={M} N={N} ln={int(ln)} epi={epi} prev={prev:.3f}: rel err {e:.2e}, recorded max {out.value:.4f} (expected {xmax:.4f}), "
              f"range flag {flag.value}", flush=True)
        assert abs(out.value - xmax) <= 1e-5 * xmax, "recorded maximum"
        worst = max(worst, e if e == e else 1.0)
    return worst


def t(M, N, mode, flags=0, iters=10, K=256):
    us = C.c_float()
    _lib.check_tools(lib.ramp_bench_gemm(M, N, K, 1, 1, mode, flags, 3, iters, C.byref(us), None))
    return us.value


if __name__ == "__main__":
    worst = 0.0
    for (M, N, ln, epi) in ((128, 768, True, 0), (293, 256, False, 3), (4096 + 77, 768, True, 0), (4096 + 77, 256, False, 0),
                            (4096 + 77, 256, False, 1), (70000, 256, False, 3), (1000, 32, False, 1), (1000, 96, False, 1)):
        worst = max(worst, check(M, N, ln, epi))
    print("worst", worst, flush=True)
    if "--bench" in sys.argv:
        for M in (393216, 196608, 98304, 49152):
            for N, fl_t, fl_g, what in ((768, 1, 0, "LN -> QKV"), (256, 2, 3, "out-proj (bias + residual)"), (256, 0, 0, "d(o)")):
                a = min(t(M, N, 8, fl_t) for _ in range(2))
                b = min(t(M, N, 3, fl_g) for _ in range(2))
                fl = 2.0 * M * N * 256
                print(f"M={M} N={N} {what:28s}: tkl {a:7.1f} us ({fl / a / 1e6:4.0f} TF)   tile kernel {b:7.1f} us ({fl / b / 1e6:4.0f} TF)", flush=True)
    if "--bench-b" in sys.argv:
        for M in (393216, 196608, 98304, 49152):
            us = C.c_float()
            best = 1e30
            for _ in range(2):
                _lib.check_tools(lib.ramp_bench_gemm(M, 256, 768, 1, 1, 9, 0, 3, 10, C.byref(us), None)); best = min(best, us.value)
            b = min(t(M, 256, 3, 0, K=768) for _ in range(2))
            fl = 2.0 * M * 256 * 768
            print(f"M={M} d(ln1) + LN1 backward: tklb {best:7.1f} us ({fl / best / 1e6:4.0f} TF)   tile kernel alone {b:7.1f} us ({fl / b / 1e6:4.0f} TF; + ln_bwd)", flush=True)
