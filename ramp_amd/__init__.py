"""ramp_amd — MI355X-native implementation of RAMP's energy-based diffusion trajectory sampler.

Python host code (reference-compatible API) over hand-written HIP kernels behind a C ABI
(include/ramp_hip.h, ramp_amd/csrc).  Import of this package never touches the GPU; the library is
loaded on first use and raises if it has not been built (no CPU fallback).
"""
__version__ = "0.1.0"
