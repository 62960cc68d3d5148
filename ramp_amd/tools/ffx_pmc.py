"""A few launches of one ramp_bench_gemm mode for rocprofv3 --pmc runs: ffx_pmc.py M mode [flags] [N K]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

M, mode = int(sys.argv[1]), int(sys.argv[2])
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N, K = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (2048, 256)
us = C.c_float()
_lib.check_tools(_lib.load_tools().ramp_bench_gemm(M, N, K, 1, 1, mode, flags, 1, 3, C.byref(us), None))
print(f"mode {mode} M {M} flags {flags}: {us.value:.1f} us")
