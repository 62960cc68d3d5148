"""Isolating experiments for atk.hip (ramp_op_ato) on small inputs."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load()


def ref(qkv, Wo, resid, L):
    M = qkv.shape[0]
    x = qkv.double().reshape(M // L, L, 3, 4, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(M, 256)
    return resid.double() + o @ Wo.double().T, o


def run(name, qkv, Wo, resid, L):
    M = qkv.shape[0]
    Y = torch.full((M, 256), float("nan"), device="cuda")
    out, flag = C.c_float(0), C.c_int32(0)
    _lib.check(lib.ramp_op_ato(_lib.ptr(qkv), _lib.ptr(Wo), None, _lib.ptr(resid), None, None, 0, L, M, 0.0, _lib.ptr(Y), C.byref(out), C.byref(flag), None), "ato")
    r, o = ref(qkv, Wo, resid, L)
    err = (Y.double() - r).abs()
    print(f"{name}: max err {err.max().item():.3e} (ref max {r.abs().max().item():.2f}), recorded amax {out.value:.4f} vs {o.abs().max().item():.4f}")
    bad = (err > 1e-4).nonzero()
    if len(bad):
        toks = sorted(set(int(b[0]) for b in bad)); cols = sorted(set(int(b[1]) for b in bad))
        print(f"   bad tokens {toks[:12]}... ({len(toks)}), bad cols {cols[:12]}... ({len(cols)})")
    return Y, r


g = torch.Generator().manual_seed(0)
L, R = 48, 4
M = L * R
rn = lambda *s: torch.randn(*s, generator=g).cuda()
eye = torch.eye(256, device="cuda")
zero = torch.zeros(M, 256, device="cuda")
# A: v depends on the feature only: o = v whatever P is
qkv = rn(M, 768); qkv[:, 512:] = torch.linspace(-1, 1, 256, device="cuda")[None, :]
run("A v=f(d), Wo=I", qkv, eye, zero, L)
# B: q = k = 0: uniform attention
qkv = rn(M, 768); qkv[:, :512] = 0
run("B q=k=0, Wo=I", qkv, eye, zero, L)
# C: everything random, Wo = I
qkv = rn(M, 768)
run("C random, Wo=I", qkv, eye, zero, L)
# D: random Wo
run("D random, Wo random", qkv, rn(256, 256) / 16, rn(M, 256), L)
