"""Bandwidth of the row kernels (GroupNorm+Mish fwd/bwd, LayerNorm fwd/bwd) at the network's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ramp_amd import _lib
lib = _lib.load()
R = 8192


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for (L, C) in [(48, 32), (24, 64), (12, 128), (6, 256), (48, 256)]:
    x = torch.randn(R, L, C, device="cuda"); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); tb = torch.randn(R, C, device="cuda")
    st = torch.empty(R, 8, 2, device="cuda")
    s = _lib.current_stream()
    tf = timeit(lambda: lib.ramp_op_groupnorm(_lib.ptr(x), _lib.ptr(g), _lib.ptr(b), _lib.ptr(tb), None, _lib.ptr(y), _lib.ptr(st), R, L, C, 1e-5, 1, s))
    tbw = timeit(lambda: lib.ramp_op_groupnorm_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(st), _lib.ptr(g), _lib.ptr(b), None, _lib.ptr(dx), R, L, C, 1, s))
    nb = x.numel() * 4
    print(f"GN R={R} L={L:2d} C={C:3d}: fwd {tf:7.1f} us {2 * nb / tf / 1e6:6.2f} TB/s   bwd {tbw:7.1f} us {3 * nb / tbw / 1e6:6.2f} TB/s", flush=True)
for L in (48, 24, 12, 6):
    n = R * L
    x = torch.randn(n, 256, device="cuda"); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x); ad = torch.randn_like(x)
    g = torch.ones(256, device="cuda"); b = torch.zeros(256, device="cuda")
    s = _lib.current_stream()
    tf = timeit(lambda: lib.ramp_op_layernorm(_lib.ptr(x), _lib.ptr(g), _lib.ptr(b), _lib.ptr(y), n, s))
    tbw = timeit(lambda: lib.ramp_op_layernorm_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(g), _lib.ptr(ad), _lib.ptr(dx), n, s))
    nb = x.numel() * 4
    print(f"LN tokens={n:7d}: fwd {tf:7.1f} us {2 * nb / tf / 1e6:6.2f} TB/s   bwd(+add) {tbw:7.1f} us {4 * nb / tbw / 1e6:6.2f} TB/s", flush=True)
