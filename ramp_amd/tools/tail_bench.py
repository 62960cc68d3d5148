import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib
lib = _lib.load_tools()
def t(M, N, K, taps, L, flags, iters=20):
    us = C.c_float()
    _lib.check_tools(lib.ramp_bench_gemm(M, N, K, taps, L, 3, flags, 3, iters, C.byref(us), None))
    return us.value
for (M, N, K, taps, fl) in [(49152, 256, 256, 1, 3), (49152, 256, 1024, 1, 3), (49152, 256, 2048, 1, 3), (49152, 256, 768, 1, 3), (49152, 768, 256, 1, 1), (49152, 1024, 256, 1, 1), (49152, 256, 256, 5, 1),
                            (98304, 256, 256, 1, 3), (98304, 256, 1024, 1, 3)]:
    L = 6 if taps > 1 else 1
    row = [f"M={M} N={N} K={K} taps={taps}:"]
    for name, f in (("auto", 0), ("128x128", 16), ("3 blocks", 32), ("128x128+3", 48)):
        try:
            us = min(t(M, N, K, taps, L, fl | f) for _ in range(3))
            row.append(f"{name} {us:.1f} us ({2.0 * M * N * K * taps / us / 1e6:.0f} TF)")
        except Exception as e:
            row.append(f"{name} failed")
    print("  ".join(row), flush=True)
