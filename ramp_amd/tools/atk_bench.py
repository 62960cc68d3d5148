"""Self-attention fused with its output projection (atk.hip, ramp_bench_gemm mode 10) against the pair it replaces (mode 11:
attn2_fwd + the token-owning out-projection), same box, the bench workload's levels."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
cases = [(48, 4096), (48, 8192), (24, 8192), (12, 8192), (6, 8192)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for L, R in cases:
    M = L * R
    row = []
    for mode in (10, 11):
        us = C.c_float(0)
        _lib.check_tools(lib.ramp_bench_gemm(M, 256, 256, 1, L, mode, 1, 3, 10, C.byref(us), None), "ramp_bench_gemm")
        row.append(us.value)
    fl = 2.0 * M * 256 * 256 + 16.0 * M * L * 64
    print(f"L={L:3d} rows={R:5d} tokens={M:7d}: fused {row[0]:8.1f} us ({fl / row[0] / 1e6:6.1f} TFLOP/s)   attn2_fwd + tkl out-proj {row[1]:8.1f} us   x{row[1] / row[0]:.2f}", flush=True)
if "--stamps" in sys.argv or os.environ.get("ATK_STAMPS"):
    for abl, what in ((0, "stamped twin"), (2, "without k / v LDS-DMA"), (4, "without ring LDS-DMA"), (6, "without either")):
        us = C.c_float(0)
        _lib.check_tools(lib.ramp_bench_gemm(48 * 8192, 256, 256, 1, 48, 10, 1 | 256 | (abl << 9), 2, 3, C.byref(us), None), "ramp_bench_gemm")
        print(f"{what}: {us.value:.1f} us", flush=True)
