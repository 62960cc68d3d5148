"""Attention kernels at the network's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ramp_amd import _lib
lib = _lib.load()
R = 8192
for L in (48, 24, 12, 6):
    qkv = torch.randn(R, L, 768, device="cuda"); o = torch.empty(R, L, 256, device="cuda")
    do = torch.randn(R, L, 256, device="cuda"); dqkv = torch.empty_like(qkv)
    s = _lib.current_stream()
    def t(fn, it=20):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / it
    tf = t(lambda: lib.ramp_op_attention(_lib.ptr(qkv), _lib.ptr(o), R, L, s))
    tb = t(lambda: lib.ramp_op_attention_bwd(_lib.ptr(qkv), _lib.ptr(do), _lib.ptr(dqkv), R, L, s))
    nb = R * L * 4
    print(f"L={L:2d}: fwd {tf:7.1f} us {nb * 1024 / tf / 1e6:5.2f} TB/s   bwd {tb:7.1f} us {nb * 1792 / tb / 1e6:5.2f} TB/s", flush=True)
