"""Long bitwise soak of the sample-owning kernels (hand-scheduled LDS-DMA + vmcnt waits): N back-to-back launches per case, every output
compared bit for bit with the first one's on the device (ramp_stress_gemm).  soak.py [launches]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

lib = _lib.load_tools()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cases = [("abl", 15, 48, 8192, 768, 0), ("abl", 15, 24, 8192, 768, 0), ("abl", 15, 12, 8192, 768, 0), ("abl", 15, 6, 8192, 768, 0), ("abl", 15, 32, 3000, 768, 0),
         ("abl", 15, 48, 1001, 768, 0), ("ato", 10, 48, 8192, 256, 1), ("ato", 10, 12, 8192, 256, 1), ("atb", 13, 48, 8192, 256, 0), ("atb", 13, 6, 8192, 256, 0)]
bad = 0
for name, mode, L, R, K, flags in cases:
    mism = C.c_int64(-1)
    _lib.check_tools(lib.ramp_stress_gemm(L * R, 256, K, 1, L, mode, flags, n, C.byref(mism), None, None), "ramp_stress_gemm")
    print(f"{name} L={L} rows={R}: {n} launches, {mism.value} differing words", flush=True)
    bad += mism.value != 0
sys.exit(1 if bad else 0)
