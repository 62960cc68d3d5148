"""Mirror of mpd/models/diffusion_models/sample_functions.py over the HIP kernels."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def apply_hard_conditioning(x, conditions):
    """x[:, t, :] = val for every (t, val) (sample_functions.py:5-10), in place, on the HIP kernel."""
    if not conditions:
        return x
    if x.device.type != "cuda":
        raise _lib.RampHipError("apply_hard_conditioning: tensor must live on a HIP device (no CPU path)")
    B, H, S = x.shape
    keys = list(conditions.keys())
    idx = (C.c_int32 * len(keys))(*[k if k >= 0 else H + k for k in keys])
    vals = []
    for k in keys:
        v = conditions[k].to(x.device, torch.float32)
        vals.append(v.unsqueeze(0).expand(B, -1) if v.dim() == 1 else v)
    val = torch.stack(vals).contiguous()
    if not x.is_contiguous():
        raise ValueError("x must be contiguous")
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().ramp_hard_cond(_lib.ptr(x), B, H, S, len(keys), idx, _lib.ptr(val),
                                              _lib.current_stream()), "ramp_hard_cond")
    return x


def extract(a, t, x_shape):
    """sample_functions.py:13-16."""
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


@torch.no_grad()
def ddpm_sample_fn(model, x, hard_conds, context, t, traj_normalized=None, obstacle_pts=None, forward_t=None,
                   compose=None, noise_std_extra_schedule_fn=None, **kwargs):
    """One DDPM step with the reference's signature (sample_functions.py:19-48).  The fused loop in
    ``ramp_sample`` is the fast path; this single-step form exists for drivers that call it directly."""
    t_single = t[0]
    if t_single < 0:
        t = torch.zeros_like(t)
    out = model.p_mean_variance(x=x, hard_conds=hard_conds, context=context, t=t, traj_normalized=traj_normalized,
                                obstacle_pts=obstacle_pts, forward_t=forward_t, compose=compose)
    model_mean = out[0]
    model_log_variance = extract(model.posterior_log_variance_clipped, t, x.shape)
    model_std = torch.exp(0.5 * model_log_variance)
    noise = torch.randn_like(x)
    noise[t == 0] = 0
    noise_std = 1.0 if noise_std_extra_schedule_fn is None else noise_std_extra_schedule_fn(t_single)
    return model_mean + model_std * noise * noise_std, None
