"""ato / abl at kernel level for a same-box A/B of two builds (RAMP_HIP_TOOLS_LIB selects the other tools library)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib
lib = _lib.load_tools()
tag = os.path.basename(os.environ.get("RAMP_HIP_TOOLS_LIB", "in-tree"))
for rnd in range(2):
    for M, L in ((393216, 48), (196608, 24), (98304, 12), (49152, 6), (131072, 32)):
        row = []
        for mode, N, K, fl, nm in ((10, 256, 256, 1, "ato"), (15, 256, 768, 0, "abl")):
            us = C.c_float()
            _lib.check_tools(lib.ramp_bench_gemm(M, N, K, 1, L, mode, fl, 30, 120, C.byref(us), None), "bench")
            row.append(f"{nm} {us.value:8.1f} us")
        print(f"{tag} round {rnd} M={M:6d} L={L:2d}: " + "   ".join(row), flush=True)
