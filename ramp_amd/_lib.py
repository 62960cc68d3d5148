"""ctypes binding of libramp_hip.so (the C ABI declared in include/ramp_hip.h).

There is no CPU fallback: if the library is missing, or fails to load, every use raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libramp_hip.so")
TOOLS_LIB_PATH = os.path.join(HERE, "lib", "libramp_hip_tools.so")

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)


class RampConfig(C.Structure):
    _fields_ = [("state_dim", C.c_int32), ("horizon", C.c_int32), ("unet_input_dim", C.c_int32),
                ("n_levels", C.c_int32), ("context_dim", C.c_int32), ("max_rows", C.c_int32),
                ("debug_taps", C.c_int32), ("gemm_mode", C.c_int32)]


class RampLaunchPlan(C.Structure):
    _fields_ = [("ff_fused_rows", C.c_int32), ("ffx_rows", C.c_int32), ("share_prefix", C.c_int32),
                ("three_blocks", C.c_int32), ("x6_pipe", C.c_int32), ("tkl_rows", C.c_int32), ("atk_rows", C.c_int32),
                ("tkc_rows", C.c_int32), ("tkw_rows", C.c_int32), ("mfma16", C.c_int32)]


class RampApfParams(C.Structure):
    _fields_ = [("cloud", C.c_void_p), ("n_points", C.c_int32), ("window", C.c_int32),
                ("window_weights_host", c_f32p), ("threshold", C.c_double), ("strength", C.c_double),
                ("passes", C.c_int32), ("reserved", C.c_int32)]


class RampSampleParams(C.Structure):
    _fields_ = [("B", C.c_int32), ("n_rp", C.c_int32), ("n_steps", C.c_int32), ("ddim", C.c_int32),
                ("w0", C.c_double), ("w1", C.c_double),
                ("t", c_i32p), ("sqrt_recip", c_f32p), ("sqrt_recipm1", c_f32p), ("coef1", c_f32p),
                ("coef2", c_f32p), ("stdv", c_f32p), ("use_noise", c_i32p), ("sqrt_a_t", c_f32p),
                ("sqrt_1m_a_t", c_f32p), ("sqrt_a_prev", c_f32p), ("dir_coef", c_f32p), ("apply_apf", c_i32p),
                ("noise_scale", c_f32p), ("clip_denoised", C.c_int32), ("predict_x0", C.c_int32),
                ("n_hard", C.c_int32), ("hard_idx_host", c_i32p), ("hard_val", C.c_void_p),
                ("apf", RampApfParams), ("use_graph", C.c_int32), ("reserved", C.c_int32),
                ("noise_mode", C.c_int32), ("reserved2", C.c_int32), ("philox_seed", C.c_uint64), ("philox_offset", C.c_uint64),
                ("philox_sample0", C.c_int64), ("philox_total", C.c_int64)]


class RampReplanParams(C.Structure):
    _fields_ = [("B", C.c_int32), ("n_rp", C.c_int32), ("n_steps", C.c_int32), ("clip_denoised", C.c_int32),
                ("w", C.c_double), ("t", c_i32p), ("sqrt_recip", c_f32p), ("sqrt_recipm1", c_f32p), ("sqrt_a_t", c_f32p),
                ("sqrt_1m_a_t", c_f32p), ("sqrt_a_prev", c_f32p), ("dir_coef", c_f32p), ("q_sqrt_a", C.c_float),
                ("q_sqrt_1m_a", C.c_float), ("n_hard", C.c_int32), ("predict_x0", C.c_int32), ("hard_idx_host", c_i32p),
                ("hard_val", C.c_void_p), ("sm_window_last", C.c_int32), ("sm_window_final", C.c_int32),
                ("sm_dt", C.c_float), ("sm_max_vel", C.c_float), ("static_pts", C.c_void_p), ("n_static", C.c_int32),
                ("n_dyn", C.c_int32), ("thr_static", C.c_double), ("thr_pred", C.c_double), ("strength_static", C.c_double),
                ("strength_pred", C.c_double), ("window_static", C.c_int32), ("n_cost", C.c_int32),
                ("cost_cloud", C.c_void_p), ("n_extra", C.c_int32), ("cost_thr", C.c_float), ("w_smooth", C.c_float),
                ("w_len", C.c_float), ("use_graph", C.c_int32)]


class RampReplanState(C.Structure):
    _fields_ = [("noise", C.c_void_p), ("x_clean", C.c_void_p), ("history", C.c_void_p), ("n_hist", C.c_int32),
                ("stepp", C.c_int32), ("dyn_pts_host", C.c_void_p), ("pursuer", C.c_float * 2), ("near", C.c_int32),
                ("reserved", C.c_int32), ("extra_pts_host", C.c_void_p)]


class RampReplanResult(C.Structure):
    _fields_ = [("n_free", C.c_int32), ("best_rank", C.c_int32), ("best_row", C.c_int32), ("fell_back", C.c_int32)]


# name -> (restype, argtypes); every symbol declared in include/ramp_hip.h
PROTOTYPES = {
    "ramp_last_error": (C.c_char_p, []),
    "ramp_version": (C.c_int, []),
    "ramp_create": (C.c_int, [C.POINTER(RampConfig), C.POINTER(C.c_void_p)]),
    "ramp_destroy": (C.c_int, [C.c_void_p]),
    "ramp_get_launch_plan": (C.c_int, [C.c_void_p, C.POINTER(RampLaunchPlan)]),
    "ramp_set_launch_plan": (C.c_int, [C.c_void_p, C.POINTER(RampLaunchPlan)]),
    "ramp_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, c_f32p, c_i64p, C.c_int32]),
    "ramp_finalize_weights": (C.c_int, [C.c_void_p]),
    "ramp_prepare_time_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ramp_time_embedding": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ramp_set_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, c_i32p, C.c_int32, C.c_void_p]),
    "ramp_encode_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "ramp_score": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                             C.c_void_p]),
    "ramp_score_mode": (C.c_int, [C.c_void_p, c_i32p]),
    "ramp_sample": (C.c_int, [C.c_void_p, C.POINTER(RampSampleParams), C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p]),
    "ramp_philox_normal": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "ramp_replan": (C.c_int, [C.c_void_p, C.POINTER(RampReplanParams), C.POINTER(RampReplanState), C.c_void_p, C.c_void_p,
                              C.c_void_p, C.POINTER(RampReplanResult), C.c_void_p]),
    "ramp_select_best": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                   C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ramp_replan_costs": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ramp_select_from_costs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p,
                                         C.c_void_p]),
    "ramp_apf": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(RampApfParams), C.c_void_p]),
    "ramp_apf_dynamic": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_double,
                                   C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ramp_hard_cond": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_i32p, C.c_void_p,
                                 C.c_void_p]),
    "ramp_traj_metrics": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ramp_waypoint_variance": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ramp_traj_costs": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_float,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ramp_cfg_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "ramp_ddim_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                   C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ramp_op_gemm": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 7 + [C.c_void_p]),
    "ramp_op_gemm_mode": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 8 + [C.c_float, c_f32p, c_i32p, C.c_void_p]),
    "ramp_op_ffx": (C.c_int, [C.c_void_p] * 8 + [C.c_int32, c_f32p, C.c_void_p, C.c_void_p, c_f32p, c_i32p, C.c_void_p]),
    "ramp_op_ffx16": (C.c_int, [C.c_void_p] * 8 + [C.c_int32, c_f32p, C.c_void_p, C.c_void_p, c_f32p, c_i32p, C.c_void_p]),
    "ramp_op_tkl": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, c_f32p, c_i32p, C.c_void_p]),
    "ramp_op_tkl16": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, c_f32p, c_i32p, C.c_void_p]),
    "ramp_op_ato": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, c_f32p, c_i32p, C.c_void_p]),
    "ramp_op_atb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "ramp_op_abl": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_float, C.c_void_p, c_f32p, c_i32p, C.c_void_p]),
    "ramp_op_tkw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 11 + [C.c_int32] * 6 + [C.c_float] + [C.c_void_p] * 4 + [c_f32p, c_i32p, C.c_void_p]),
    "ramp_op_tklb": (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_float, C.c_void_p, c_f32p, c_i32p, C.c_void_p]),
    "ramp_op_groupnorm": (C.c_int, [C.c_void_p] * 7 + [C.c_int32] * 3 + [C.c_float, C.c_int32, C.c_void_p]),
    "ramp_op_groupnorm_bwd": (C.c_int, [C.c_void_p] * 7 + [C.c_int32] * 4 + [C.c_void_p]),
    "ramp_op_layernorm": (C.c_int, [C.c_void_p] * 4 + [C.c_int32, C.c_void_p]),
    "ramp_op_layernorm_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_void_p]),
    "ramp_op_geglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "ramp_op_geglu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "ramp_op_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "ramp_op_attention_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "ramp_debug_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int64, c_i64p, C.c_void_p]),
    "ramp_profile": (C.c_int, [C.c_void_p, C.c_int32]),
    "ramp_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), c_i64p]),
    "ramp_profile_read_kernels": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), c_i64p]),
    "ramp_set_fallback": (C.c_int, [C.c_void_p, C.c_int32]),
    "ramp_set_calibration_reuse": (C.c_int, [C.c_void_p, C.c_int32]),
    "ramp_range_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "ramp_range_trip": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ramp_workspace_bytes": (C.c_int, [C.c_void_p, c_i64p]),
    "ramp_launch_count": (C.c_int, [C.c_void_p, c_i64p]),
}

# Diagnostics, exported by ramp_amd/lib/libramp_hip_tools.so only (csrc/bench.hip: the product library carries no harness code)
TOOL_PROTOTYPES = {
    "ramp_bench_gemm": (C.c_int, [C.c_int32] * 9 + [c_f32p, C.c_void_p]),
    "ramp_stress_gemm": (C.c_int, [C.c_int32] * 8 + [c_i64p, c_f32p, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None
_tools: Optional[C.CDLL] = None


class RampHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libramp_hip.so and bind every prototype.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64; the process must hold ONE HIP runtime (torch's
    # streams and allocations are handed to this library), so torch is always imported first and this library's
    # NEEDED libamdhip64.so.7 then resolves to the copy already mapped.  Loaded the other way round the second
    # runtime finds no device.
    import torch  # noqa: F401
    # RAMP_HIP_LIB: another build of the same library for a same-box A/B (ramp_amd/tools/ab_libs.sh); the in-tree one is never
    # overwritten, so later runs cannot pick up a stale alternate by accident
    path = os.environ.get("RAMP_HIP_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise RampHipError(f"{path} not found: build it with `python -m ramp_amd.build` "
                           "(hipcc --offload-arch=gfx950). The RAMP sampler has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def load_tools() -> C.CDLL:
    """The tools library: every object of the product library plus the micro-benchmark / stress harness (ramp_bench_gemm,
    ramp_stress_gemm; csrc/bench.hip).  For tests/ and ramp_amd/tools/ only -- nothing in the product path imports it."""
    global _tools
    if _tools is not None:
        return _tools
    import torch  # noqa: F401  (one HIP runtime per process: see load())
    path = os.environ.get("RAMP_HIP_TOOLS_LIB") or TOOLS_LIB_PATH
    if not os.path.exists(path):
        raise RampHipError(f"{path} not found: build it with `python -m ramp_amd.build`")
    lib = C.CDLL(path)
    for name, (res, args) in list(PROTOTYPES.items()) + list(TOOL_PROTOTYPES.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _tools = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ramp_last_error()
        raise RampHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def check_tools(rc: int, what: str = "") -> None:
    """check() for a call made through load_tools() (each library keeps its own thread-local error message)."""
    if rc != 0:
        msg = load_tools().ramp_last_error()
        raise RampHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


GEMM_MODES = {"fp32": 0, "bf16x6": 1, "bf16x6-lds": 2, "fp16x3": 3, "fp16x3-tkc": 5}       # ramp_op_gemm_mode


def op_gemm(A, W, bias, resid, out, M, N, K, taps, shift0, step, L, mode="fp32", a_absmax_prev=0.0):
    """ramp_op_gemm_mode on torch tensors; returns (recorded max|A|, range flag) (zeros outside fp16x3)."""
    amax, flag = C.c_float(0.0), C.c_int32(0)
    check(load().ramp_op_gemm_mode(ptr(A), ptr(W), ptr(bias), ptr(resid), ptr(out), M, N, K, taps, shift0, step, L,
                                   GEMM_MODES[mode], float(a_absmax_prev), C.byref(amax), C.byref(flag),
                                   current_stream()), "ramp_op_gemm_mode")
    return amax.value, flag.value


def ptr(t) -> Optional[int]:
    """Device (or host) address of a contiguous torch tensor, None for None."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()


def current_stream() -> Optional[int]:
    import torch
    return torch.cuda.current_stream().cuda_stream
