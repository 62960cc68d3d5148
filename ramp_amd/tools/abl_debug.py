"""Where does abl_kernel (atl.hip, through ramp_op_abl) differ from float64 autograd?  abl_debug.py L R: error per 16-token group and per
16-feature block of the output, the recorded maximum against max |d(qkv)|, and atb_kernel on the same operands for comparison."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from ramp_amd import _lib
L, R = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "full"
M = R * L
gen = torch.Generator(device="cpu").manual_seed(91 * L + R)
z = (torch.randn(M, 256, generator=gen) * 1.5 + 0.3)
ln_g = 1.0 + 0.2 * torch.randn(256, generator=gen)
ln_b = 0.1 * torch.randn(256, generator=gen)
Wqkv = torch.randn(768, 256, generator=gen) / 16.0
dout = torch.randn(M, 256, generator=gen)
add = torch.randn(M, 256, generator=gen)
zd = z.double().clone().requires_grad_(True)
ln = torch.nn.functional.layer_norm(zd, (256,), ln_g.double(), ln_b.double(), 1e-5)
qkv64 = ln @ Wqkv.double().t()
qkv64.retain_grad()
y = qkv64.reshape(R, L, 3, 4, 64)
q, k, v = y[:, :, 0].transpose(1, 2), y[:, :, 1].transpose(1, 2), y[:, :, 2].transpose(1, 2)
o = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v).transpose(1, 2).reshape(M, 256)
o.backward(dout.double())
ref = (zd.grad + add.double()).numpy()
dq = qkv64.grad
print("true max", float(dq.abs().max()), "per part", [float(dq[:, i*256:(i+1)*256].abs().max()) for i in range(3)])
qkv = qkv64.detach().float().cuda()
Wb = Wqkv.t().contiguous().cuda()
lib = _lib.load()
got = torch.full((M, 256), float("nan"), device="cuda")
amax, flag = C.c_float(0), C.c_int32(0)
args = [dout.cuda(), z.cuda(), ln_g.cuda(), add.cuda()]
_lib.check(lib.ramp_op_abl(_lib.ptr(qkv), _lib.ptr(args[0]), _lib.ptr(Wb), _lib.ptr(args[1]), _lib.ptr(args[2]), _lib.ptr(args[3]),
                           M, L, 0.0, _lib.ptr(got), C.byref(amax), C.byref(flag), None), "ramp_op_abl")
got = got.cpu().numpy()
print("amax", amax.value, "flag", flag.value)
err = np.abs(got - ref)
print("rel", err.max() / np.abs(ref).max(), "nan", np.isnan(got).sum())
print("per 16-token group max err:", [float(err[i:i+16].max()) for i in range(0, min(M, 192), 16)])
print("per 16-feature block max err:", [round(float(err[:, i:i+16].max()), 4) for i in range(0, 256, 16)])
# reference via the separate kernels: atb -> dqkv ; compare got vs tklb(dqkv)
dqkv = torch.empty(M, 768, device="cuda")
_lib.check(lib.ramp_op_atb(_lib.ptr(qkv), _lib.ptr(args[0]), _lib.ptr(dqkv), M, L, None), "atb")
print("atb rel", float((dqkv.double().cpu() - dq).abs().max() / dq.abs().max()))
