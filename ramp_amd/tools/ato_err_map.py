"""Where ramp_op_ato's error against float64 sits: max |err| per (token group of 16 within a sample-owning wave tile, 16-feature block).  Diagnostic."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib
L, R = 48, 9
M = R * L
gen = torch.Generator(device="cpu").manual_seed(1000 * L + R)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).cuda()
qkv = r(M, 768, sc=1.5)
qkv[:, 512:] = qkv[:, 512:] * 0.3 + 0.1
Wo, bias, resid = r(256, 256, sc=1 / 16), r(256, sc=0.3), r(M, 256)
x = qkv.double().reshape(M // L, L, 3, 4, 64)
q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
o = (p @ v).transpose(1, 2).reshape(M, 256)
ref = (resid.double() + o @ Wo.double().T + bias.double()).cpu().numpy()
Y = torch.empty(M, 256, device="cuda")
out, flag = C.c_float(0), C.c_int32(0)
_lib.check(_lib.load().ramp_op_ato(_lib.ptr(qkv), _lib.ptr(Wo), _lib.ptr(bias), _lib.ptr(resid), None, None, 0, L, M, 0.0, _lib.ptr(Y), C.byref(out), C.byref(flag), None), "ato")
e = np.abs(Y.double().cpu().numpy() - ref)
print("max err", e.max(), "scale", np.abs(ref).max())
tok_err = e.max(1)
print("per token (first 2 samples):", np.array2string(tok_err[:96], precision=1, max_line_width=200))
print("per 16-feature block:", np.array2string(e.reshape(M, 16, 16).max(axis=(0, 2)), precision=1, max_line_width=200))
print("per sample:", np.array2string(e.reshape(R, -1).max(1), precision=1))
# which head's contribution is off?  project the error back through Wo^-1 is overkill: compare o instead via least squares
