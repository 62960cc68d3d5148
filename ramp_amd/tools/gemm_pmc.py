"""One GEMM shape repeated a few times (for rocprofv3 --pmc runs): gemm_pmc.py M N K [mode] [flags]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ramp_amd import _lib

M, N, K = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "fp16x3"
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
us = C.c_float()
_lib.check_tools(_lib.load_tools().ramp_bench_gemm(M, N, K, 1, 1, _lib.GEMM_MODES[mode], flags, 2, 5, C.byref(us), None))
print(f"{mode} {M}x{N}x{K} flags {flags}: {us.value:.1f} us")
