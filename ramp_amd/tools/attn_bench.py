"""Attention forward / backward kernels alone (R rows x 4 heads x L tokens), HIP-event timed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from ramp_amd import _lib  # noqa: E402

lib = _lib.load()
for (R, L) in [(8192, 48), (4096, 48), (8192, 24), (8192, 12), (8192, 6), (8192, 64), (8192, 32), (8192, 16), (8192, 8)]:
    qkv = torch.randn(R * L, 768, device="cuda")
    do = torch.randn(R * L, 256, device="cuda")
    o = torch.empty(R * L, 256, device="cuda")
    dqkv = torch.empty(R * L, 768, device="cuda")
    s = _lib.current_stream()

    def timeit(fn, n=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    tf = timeit(lambda: _lib.check(lib.ramp_op_attention(_lib.ptr(qkv), _lib.ptr(o), R, L, s)))
    tb = timeit(lambda: _lib.check(lib.ramp_op_attention_bwd(_lib.ptr(qkv), _lib.ptr(do), _lib.ptr(dqkv), R, L, s)))
    bf = R * L * (768 + 256) * 4; bb = R * L * (768 + 256 + 768) * 4
    print(f"R={R} L={L}: fwd {tf:.0f} us ({bf / tf / 1e6:.2f} TB/s)  bwd {tb:.0f} us ({bb / tb / 1e6:.2f} TB/s)", flush=True)
