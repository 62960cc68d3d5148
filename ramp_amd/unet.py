"""Host-side mirror of ``mpd.models.TemporalUnetInference`` over the HIP C ABI.

Same constructor kwargs, same ``forward`` / ``forward_no_energy`` / ``reset_cache`` signatures,
same state_dict key names as the reference (UnetInference.py:40-230), so it drops in behind
``scripts/inference/inference_static.py``-style drivers.  All arithmetic of the U-Net runs in
libramp_hip.so (there is no PyTorch fallback); PyTorch is used for device memory, streams and
the once-per-scene encoder only.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from . import _lib
from .scene_encoder import ObstacleEncoder, ObstacleEncoderSet
from .spec import UNET_DIM_MULTS, make_unet_spec, unet_param_shapes  # noqa: F401  (re-export)


class TemporalUnetInference(nn.Module):
    """Energy-gradient temporal U-Net: ``forward`` returns eps = d/dx 0.5*||f(x, t, scene)||^2."""

    def __init__(self, n_support_points=None, state_dim=None, unet_input_dim=32, dim_mults=(1, 2, 4, 8),
                 time_emb_dim=32, self_attention=False, conditioning_embed_dim=4, conditioning_type='attention',
                 attention_num_heads=4, attention_dim_head=64, obstacle_3d=False, max_rows: int = 8192,
                 debug_taps: bool = False, gemm_mode: str = "default", launch_plan: Optional[dict] = None, **kwargs):
        super().__init__()
        if self_attention:
            raise NotImplementedError("self_attention=True (LinearAttention) is never used by the reference drivers")
        if conditioning_type != 'attention':
            raise NotImplementedError("only conditioning_type='attention' is on the sampler hot path")
        if time_emb_dim != 32 or attention_num_heads != 4 or attention_dim_head != 64:
            raise NotImplementedError("kernels are built for time_emb_dim=32 and 4x64 attention heads")
        self.state_dim = state_dim
        self.n_support_points = n_support_points
        self.obstacle_3d = obstacle_3d
        self.energy_mode = True
        self.cfg_batch = not obstacle_3d                      # UnetInference.py:73
        self.context_dim = 256 if obstacle_3d else 320
        self.conditioning_type = conditioning_type
        self.enable_caching = True
        self.spec = make_unet_spec(state_dim, n_support_points, unet_input_dim, tuple(dim_mults), obstacle_3d)
        self.scene_encoder = ObstacleEncoder() if obstacle_3d else ObstacleEncoderSet()
        self.max_rows = int(max_rows)
        self.debug_taps = bool(debug_taps)
        self.gemm_mode = {"default": 0, "fp32": 1, "bf16x6": 2, "fp16x3": 3}[gemm_mode]
        # performance knobs of the HIP engine (ramp_launch_plan: ff_fused_rows, ffx_rows, tkl_rows, share_prefix, three_blocks,
        # x6_pipe); None keeps the library defaults.  Every plan meets the same parity bar.
        self.launch_plan = dict(launch_plan or {})
        self._unet_keys = [k for k in unet_param_shapes(self.spec, with_scene_encoder=False)]
        self._unet_shapes = unet_param_shapes(self.spec, with_scene_encoder=False)
        self._weights: "OrderedDict[str, torch.Tensor]" = OrderedDict()   # host fp32 copies (checkpoint truth)
        self._ctx: Optional[C.c_void_p] = None
        self._T_table = 0
        self._scene_key = None
        self._scene_ref = None
        self.cached_scene_latents = None
        self.cached_batch_size = None

    # ------------------------------------------------------------------ state dict
    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        sd = OrderedDict() if destination is None else destination
        for k, v in self.scene_encoder.state_dict().items():
            sd[prefix + "scene_encoder." + k] = v
        for k, v in self._weights.items():
            sd[prefix + k] = v
        return sd

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        # consume the U-Net tensors under `prefix`; scene_encoder.* is handled by the child module
        for k in self._unet_keys:
            full = prefix + k
            if full in state_dict:
                t = state_dict[full].detach().to("cpu", torch.float32).contiguous()
                if tuple(t.shape) != tuple(self._unet_shapes[k]):
                    error_msgs.append(f"size mismatch for {full}: {tuple(t.shape)} vs {tuple(self._unet_shapes[k])}")
                    continue
                self._weights[k] = t.clone()
            elif strict:
                missing_keys.append(full)
        if strict:
            known = set(prefix + k for k in self._unet_keys)
            for k in state_dict:
                if k.startswith(prefix) and not k.startswith(prefix + "scene_encoder.") and k not in known:
                    unexpected_keys.append(k)
        self._destroy_ctx()

    # ------------------------------------------------------------------ context management
    def _destroy_ctx(self):
        if self._ctx is not None:
            _lib.load().ramp_destroy(self._ctx)
            self._ctx = None
            self._T_table = 0
            self._scene_key = None
            self._scene_ref = None

    def __del__(self):
        try:
            self._destroy_ctx()
        except Exception:
            pass

    def _device(self) -> torch.device:
        return next(self.scene_encoder.parameters()).device

    def ctx(self) -> C.c_void_p:
        """Create (lazily) the HIP context and upload + pack the weights."""
        if self._ctx is not None:
            return self._ctx
        missing = [k for k in self._unet_keys if k not in self._weights]
        if missing:
            raise RuntimeError(f"{len(missing)} U-Net tensors not loaded (first: {missing[0]}); call load_state_dict")
        dev = self._device()
        if dev.type != "cuda":
            raise _lib.RampHipError("TemporalUnetInference must be moved to a HIP device (.to('cuda')): "
                                    "the RAMP sampler has no CPU path")
        lib = _lib.load()
        with torch.cuda.device(dev):
            cfg = _lib.RampConfig(self.state_dim, self.n_support_points, self.spec.unet_input_dim,
                                  len(self.spec.dim_mults), self.context_dim, self.max_rows, int(self.debug_taps),
                                  self.gemm_mode)
            h = C.c_void_p()
            _lib.check(lib.ramp_create(C.byref(cfg), C.byref(h)), "ramp_create")
            if self.launch_plan:
                self._apply_plan(h, self.launch_plan)
            for k in self._unet_keys:
                w = self._weights[k]
                shape = (C.c_int64 * w.dim())(*w.shape)
                _lib.check(lib.ramp_load_weight(h, k.encode(), C.cast(w.data_ptr(), _lib.c_f32p), shape, w.dim()),
                           f"ramp_load_weight({k})")
            for k, v in self.scene_encoder.state_dict().items():          # float tensors incl. BN running stats
                if not torch.is_floating_point(v):
                    continue
                w = v.detach().to("cpu", torch.float32).contiguous()
                shape = (C.c_int64 * max(w.dim(), 1))(*(w.shape if w.dim() else (1,)))
                _lib.check(lib.ramp_load_weight(h, ("scene_encoder." + k).encode(), C.cast(w.data_ptr(), _lib.c_f32p),
                                                shape, max(w.dim(), 1)), f"ramp_load_weight(scene_encoder.{k})")
            _lib.check(lib.ramp_finalize_weights(h), "ramp_finalize_weights")
        self._ctx = h
        return h

    @staticmethod
    def _apply_plan(h, kw: dict):
        lib = _lib.load()
        plan = _lib.RampLaunchPlan()
        _lib.check(lib.ramp_get_launch_plan(h, C.byref(plan)), "ramp_get_launch_plan")
        for k, v in kw.items():
            if k not in ("ff_fused_rows", "ffx_rows", "share_prefix", "three_blocks", "x6_pipe", "tkl_rows", "atk_rows", "tkc_rows", "tkw_rows", "mfma16"):
                raise KeyError(f"unknown launch-plan field {k!r}")
            setattr(plan, k, int(v))
        _lib.check(lib.ramp_set_launch_plan(h, C.byref(plan)), "ramp_set_launch_plan")

    def set_launch_plan(self, **kw):
        """Change performance knobs of the engine (see ``include/ramp_hip.h``, ramp_launch_plan); takes effect at the next
        evaluation (captured graphs and kept fp16x3 calibrations of the old plan are dropped)."""
        self.launch_plan.update(kw)
        if self._ctx is not None:
            with torch.cuda.device(self._device()):
                self._apply_plan(self._ctx, kw)

    def get_launch_plan(self) -> dict:
        plan = _lib.RampLaunchPlan()
        _lib.check(_lib.load().ramp_get_launch_plan(self.ctx(), C.byref(plan)), "ramp_get_launch_plan")
        return {k: getattr(plan, k) for k in ("ff_fused_rows", "ffx_rows", "share_prefix", "three_blocks", "x6_pipe", "tkl_rows", "atk_rows", "tkc_rows", "tkw_rows", "mfma16")}

    def prepare_time_table(self, T: int):
        if T > self._T_table:
            with torch.cuda.device(self._device()):
                _lib.check(_lib.load().ramp_prepare_time_table(self.ctx(), int(T), _lib.current_stream()),
                           "ramp_prepare_time_table")
            self._T_table = int(T)

    def time_embedding(self, t: int) -> torch.Tensor:
        """The TimeEncoder output for timestep t (layers.py:233-259) as the engine's time table holds it, (32,)."""
        self.prepare_time_table(int(t) + 1)
        out = torch.empty(32, device=self._device(), dtype=torch.float32)
        with torch.cuda.device(self._device()):
            _lib.check(_lib.load().ramp_time_embedding(self.ctx(), int(t), _lib.ptr(out), _lib.current_stream()),
                       "ramp_time_embedding")
        return out

    # ------------------------------------------------------------------ scene
    @torch.no_grad()
    def encode_scene(self, cloud: torch.Tensor) -> torch.Tensor:
        """cloud (No,Np,D) or (n_scenes,No,Np,D) -> latents (n_scenes, context_dim), on the HIP kernels
        (ramp_encode_scene).  ``self.scene_encoder`` (torch) only holds the parameters / state_dict."""
        if cloud.dim() == 3:
            cloud = cloud.unsqueeze(0)
        cloud = cloud.to(self._device(), torch.float32).contiguous()
        out = torch.empty((cloud.shape[0], self.context_dim), device=self._device(), dtype=torch.float32)
        with torch.cuda.device(self._device()):
            for i in range(cloud.shape[0]):
                _lib.check(_lib.load().ramp_encode_scene(self.ctx(), _lib.ptr(cloud[i]), cloud.shape[1], cloud.shape[2],
                                                         cloud.shape[3], out[i].data_ptr(), _lib.current_stream()),
                           "ramp_encode_scene")
        return out

    def set_scene(self, latents: torch.Tensor, row_pattern: List[int]):
        """latents (n_variants, ctx) with all-zero rows for unconditional variants; row r of the network
        uses variant row_pattern[r % len(row_pattern)]."""
        lat = latents.to(self._device(), torch.float32).contiguous()
        self._scene_key = None
        self._scene_ref = None
        pat = (C.c_int32 * len(row_pattern))(*row_pattern)
        with torch.cuda.device(self._device()):
            _lib.check(_lib.load().ramp_set_scene(self.ctx(), _lib.ptr(lat), lat.shape[0], pat, len(row_pattern),
                                                  _lib.current_stream()), "ramp_set_scene")
        self.cached_scene_latents = lat

    def cache_scene_encoding(self, obstacle_pts: torch.Tensor, compose: bool = False):
        """Reference semantics (UnetInference.py:146-156): recompute only when the cache is empty or the
        batch size changed; ``obstacle_pts`` (N,No,Np,D) holds one cloud per network row."""
        n = obstacle_pts.shape[0]
        if self.enable_caching and self.cached_scene_latents is not None and self.cached_batch_size == n:
            return self.cached_scene_latents
        zero = torch.zeros(1, self.context_dim, device=self._device())
        if compose:                       # rows [scene A, scene B, uncond] repeating (UnetInference.py:190-191)
            lat = torch.cat([self.encode_scene(obstacle_pts[0:2]), zero])
            pattern = [0, 1, 2]
        elif self.cfg_batch:              # odd rows unconditional (UnetInference.py:192-195)
            lat = torch.cat([self.encode_scene(obstacle_pts[0]), zero])
            pattern = [0, 1]
        else:                             # 3-D: row 1 only (UnetInference.py:196-197)
            if n > 64:
                raise ValueError("3-D reference masking (row 1 unconditional) supports at most 64 rows here; "
                                 "use the batched sampler API instead")
            lat = torch.cat([self.encode_scene(obstacle_pts[0]), zero])
            pattern = [0, 1] + [0] * (n - 2) if n >= 2 else [0]
        self.set_scene(lat, pattern)
        self.cached_batch_size = n
        return self.cached_scene_latents

    def reset_cache(self):
        self.cached_scene_latents = None
        self.cached_batch_size = None
        self.invalidate_scene()             # the samplers' content-keyed cache too: forces a re-encode

    def invalidate_scene(self):
        """Forget the scene the samplers encoded last.  The samplers skip the comparison of the cloud's CONTENT when they are
        handed the very tensor object they saw last time with an unchanged autograd version counter; a write that does not
        bump that counter (``x.data.copy_()``, a raw-pointer kernel, a DLPack / externally shared buffer) must be followed by
        this call (or ``reset_cache()``), otherwise the previous scene encoding stays in use."""
        self._scene_key = None
        self._scene_ref = None
        self._scene_ident = None
        self._scene_src = None

    # ------------------------------------------------------------------ forward
    def _run(self, x, time, obstacle_pts, compose, want_f, want_eps):
        if obstacle_pts is None:
            raise ValueError("obstacle_pts is required (the reference dereferences it unconditionally)")
        x = x.detach().to(self._device(), torch.float32).contiguous()
        if torch.is_tensor(time):
            tv = time.reshape(-1)
            # the time-conditioning tables are per step, not per row: the samplers always pass make_timesteps'
            # batch-uniform vector (diffusion_model_static.py:16-18); per-row timesteps are refused, not ignored
            if tv.numel() > 1 and not bool((tv == tv[0]).all()):
                raise ValueError("per-row timesteps are not supported: `time` must be uniform over the batch")
            t = int(tv[0])
        else:
            t = int(time)
        self.prepare_time_table(max(t + 1, self._T_table, 1))
        self.cache_scene_encoding(obstacle_pts, compose)
        n = x.shape[0]
        f = torch.empty_like(x) if want_f else None
        eps = torch.empty_like(x) if want_eps else None
        with torch.cuda.device(self._device()):
            _lib.check(_lib.load().ramp_score(self.ctx(), _lib.ptr(x), n, 1, t, _lib.ptr(f), _lib.ptr(eps),
                                              _lib.current_stream()), "ramp_score")
        return f, eps

    def forward(self, x, time, context, x_start=None, obstacle_pts=None, forward_t=None, compose=False):
        """eps (N,H,S).  ``context``, ``x_start``, ``forward_t`` are accepted and ignored like the reference."""
        if not self.energy_mode:
            return self.forward_no_energy(x, time, x_start=x_start, obstacle_pts=obstacle_pts, forward_t=forward_t,
                                          compose=compose)
        return self._run(x, time, obstacle_pts, compose, False, True)[1]

    def forward_no_energy(self, x, time, x_start=None, obstacle_pts=None, forward_t=None, compose=False):
        return self._run(x, time, obstacle_pts, compose, True, False)[0]

    # ------------------------------------------------------------------ debug taps
    def debug_read(self, kind: str, module: str, shape) -> torch.Tensor:
        out = torch.empty(shape, device=self._device(), dtype=torch.float32)
        n = C.c_int64()
        _lib.check(_lib.load().ramp_debug_read(self.ctx(), kind.encode(), module.encode(), _lib.ptr(out), out.numel(),
                                               C.byref(n), _lib.current_stream()), "ramp_debug_read")
        if n.value != out.numel():
            raise RuntimeError(f"tap {kind}/{module}: got {n.value} floats, expected {out.numel()}")
        return out

    def score_mode(self) -> str:
        """Arithmetic of the last single evaluation (forward / p_mean_variance): 'fp32', 'bf16x6' or 'fp16x3'."""
        n = C.c_int32()
        _lib.check(_lib.load().ramp_score_mode(self.ctx(), C.byref(n)))
        return {0: "fp32", 1: "bf16x6", 2: "fp16x3"}[n.value]

    def set_calibration_reuse(self, on: bool) -> None:
        """fp16x3: whether a sampling job may start from the operand maxima of the previous job of the same shape
        (default) or calibrates itself every time (ramp_set_calibration_reuse, include/ramp_hip.h)."""
        _lib.check(_lib.load().ramp_set_calibration_reuse(self.ctx(), int(bool(on))), "ramp_set_calibration_reuse")

    def workspace_bytes(self) -> int:
        n = C.c_int64()
        _lib.check(_lib.load().ramp_workspace_bytes(self.ctx(), C.byref(n)))
        return n.value

    def launch_count(self) -> int:
        n = C.c_int64()
        _lib.check(_lib.load().ramp_launch_count(self.ctx(), C.byref(n)))
        return n.value


def load_numpy_state_dict(model: nn.Module, sd: Dict[str, np.ndarray], prefix: str = ""):
    """Helper for synthetic weights (ramp_amd.synth): numpy dict -> load_state_dict."""
    model.load_state_dict({prefix + k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return model
