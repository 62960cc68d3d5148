"""PyTorch-CPU eager baseline of the sampler hot path — TEST / BENCH INFRASTRUCTURE ONLY.

BASELINE.json's north_star asks for "the reference PyTorch-CPU sampler ... timed on the GPU box's own host cores".
The reference sources never travel to the GPU box, so this file defines the SAME architecture and loop in this repo's own
words: a functional torch implementation over the flat state dict (reference key names), ATen fp32 CPU kernels,
``torch.autograd.grad`` for the energy gradient exactly like the reference's EnergyGradFunction
(UnetInference.py:19-37) — i.e. the same arithmetic engine and the same amount of work per trajectory as the reference's
eager CPU path, without its module tree.  It is pinned against the reference's own outputs
(tests/test_oracle_vs_golden.py::test_torch_cpu_baseline_*: ``unet2d_h48.npz`` forward / eps, ``chain_ddpm_plain.npz``)
and is imported only by tests/ and bench.py's ``cpu_baseline`` leg, never by the product path.

Layer citations: TimeEncoder layers.py:233-259; ResidualTemporalBlock / Conv1dBlock layers.py:280-361;
Downsample1d / Upsample1d layers.py:262-277; SpatialTransformer / BasicTransformerBlock / CrossAttention / GEGLU
layers_attention_mini.py:38-45, 83-202; network wiring UnetInference.py:176-224; CFG + x0 + posterior
diffusion_model_static.py:149-186; DDPM step sample_functions.py:19-48.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F


def _groups(c: int) -> int:
    """layers.py:429-435 (target 8 groups for every channel count used here)."""
    return 8 if c % 8 == 0 else 1


class TorchCpuScoreNet:
    """f(x, t, latent) with channels-first Conv1d like the reference; eps through autograd."""

    def __init__(self, sd: Dict[str, np.ndarray], state_dim: int, horizon: int, n_levels: int = 4):
        self.p = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()
                  if np.asarray(v).dtype.kind == "f" and not k.startswith("scene_encoder.")}
        self.S, self.H, self.nl = state_dim, horizon, n_levels

    def _temb(self, t: torch.Tensor) -> torch.Tensor:
        p = self.p
        half = 16
        freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
        e = t.float()[:, None] * freq[None, :]
        e = torch.cat([e.sin(), e.cos()], dim=-1)
        h = F.mish(F.linear(e, p["time_mlp.encoder.1.weight"], p["time_mlp.encoder.1.bias"]))
        return F.linear(h, p["time_mlp.encoder.3.weight"], p["time_mlp.encoder.3.bias"])

    def _rtb(self, n: str, x: torch.Tensor, temb: torch.Tensor) -> torch.Tensor:
        p = self.p
        w1 = p[f"{n}.blocks.0.block.0.weight"]
        g = _groups(w1.shape[0])
        h = F.conv1d(x, w1, p[f"{n}.blocks.0.block.0.bias"], padding=2)
        h = F.mish(F.group_norm(h, g, p[f"{n}.blocks.0.block.2.weight"], p[f"{n}.blocks.0.block.2.bias"], 1e-5))
        h = h + F.linear(F.silu(temb), p[f"{n}.cond_mlp.1.weight"], p[f"{n}.cond_mlp.1.bias"])[:, :, None]
        h = F.conv1d(h, p[f"{n}.blocks.1.block.0.weight"], p[f"{n}.blocks.1.block.0.bias"], padding=2)
        h = F.mish(F.group_norm(h, g, p[f"{n}.blocks.1.block.2.weight"], p[f"{n}.blocks.1.block.2.bias"], 1e-5))
        if f"{n}.residual_conv.weight" in p:
            x = F.conv1d(x, p[f"{n}.residual_conv.weight"], p[f"{n}.residual_conv.bias"])
        return h + x

    def _st(self, n: str, x: torch.Tensor, lat: torch.Tensor) -> torch.Tensor:
        p = self.p
        N, C, L = x.shape
        z = F.group_norm(x, _groups(C), p[f"{n}.norm.weight"], p[f"{n}.norm.bias"], 1e-6)
        z = F.conv1d(z, p[f"{n}.proj_in.weight"], p[f"{n}.proj_in.bias"]).transpose(1, 2)          # (N, L, 256)
        for b in range(2):
            t = f"{n}.transformer_blocks.{b}"
            ln = F.layer_norm(z, (256,), p[f"{t}.norm1.weight"], p[f"{t}.norm1.bias"])
            q, k, v = (F.linear(ln, p[f"{t}.attn1.to_{c}.weight"]).view(N, L, 4, 64).transpose(1, 2) for c in "qkv")
            a = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1) @ v
            a = a.transpose(1, 2).reshape(N, L, 256)
            z = F.linear(a, p[f"{t}.attn1.to_out.0.weight"], p[f"{t}.attn1.to_out.0.bias"]) + z
            # attn2: ONE context token, so softmax == 1 and the block adds to_out(to_v(ctx)) to every token; the reference
            # still evaluates norm2 / to_q / to_k and a 1-key softmax, which is where its extra 10 % of FLOPs go
            ln2 = F.layer_norm(z, (256,), p[f"{t}.norm2.weight"], p[f"{t}.norm2.bias"])
            q2 = F.linear(ln2, p[f"{t}.attn2.to_q.weight"]).view(N, L, 4, 64).transpose(1, 2)
            k2 = F.linear(lat, p[f"{t}.attn2.to_k.weight"]).view(N, 1, 4, 64).transpose(1, 2)
            v2 = F.linear(lat, p[f"{t}.attn2.to_v.weight"]).view(N, 1, 4, 64).transpose(1, 2)
            a2 = (torch.softmax((q2 @ k2.transpose(-1, -2)) * 0.125, dim=-1) @ v2).transpose(1, 2).reshape(N, L, 256)
            z = F.linear(a2, p[f"{t}.attn2.to_out.0.weight"], p[f"{t}.attn2.to_out.0.bias"]) + z
            ln = F.layer_norm(z, (256,), p[f"{t}.norm3.weight"], p[f"{t}.norm3.bias"])
            ag = F.linear(ln, p[f"{t}.ff.net.0.proj.weight"], p[f"{t}.ff.net.0.proj.bias"])
            aa, gg = ag.chunk(2, dim=-1)
            z = F.linear(aa * F.gelu(gg), p[f"{t}.ff.net.2.weight"], p[f"{t}.ff.net.2.bias"]) + z
        return F.conv1d(z.transpose(1, 2), p[f"{n}.proj_out.weight"], p[f"{n}.proj_out.bias"]) + x

    def f(self, x: torch.Tensor, t: torch.Tensor, lat: torch.Tensor) -> torch.Tensor:
        """x (N,H,S), t (N,) long, lat (N,ctx) with unconditional rows zeroed -> (N,H,S)."""
        p, nl = self.p, self.nl
        temb = self._temb(t)
        h = x.transpose(1, 2)
        skips = []
        for k in range(nl):
            h = self._rtb(f"downs.{k}.0", h, temb)
            h = self._rtb(f"downs.{k}.1", h, temb)
            h = self._st(f"downs.{k}.3", h, lat)
            skips.append(h)
            if k < nl - 1:
                h = F.conv1d(h, p[f"downs.{k}.4.conv.weight"], p[f"downs.{k}.4.conv.bias"], stride=2, padding=1)
        h = self._rtb("mid_block1", h, temb)
        h = self._st("mid_attention", h, lat)
        h = self._rtb("mid_block2", h, temb)
        for k in range(nl - 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = self._rtb(f"ups.{k}.0", h, temb)
            h = self._rtb(f"ups.{k}.1", h, temb)
            h = self._st(f"ups.{k}.3", h, lat)
            h = F.conv_transpose1d(h, p[f"ups.{k}.4.conv.weight"], p[f"ups.{k}.4.conv.bias"], stride=2, padding=1)
        h = F.conv1d(h, p["final_conv.0.block.0.weight"], p["final_conv.0.block.0.bias"], padding=2)
        h = F.mish(F.group_norm(h, _groups(h.shape[1]), p["final_conv.0.block.2.weight"], p["final_conv.0.block.2.bias"], 1e-5))
        return F.conv1d(h, p["final_conv.1.weight"], p["final_conv.1.bias"]).transpose(1, 2)

    def score(self, x: torch.Tensor, t: torch.Tensor, lat: torch.Tensor) -> torch.Tensor:
        with torch.enable_grad():
            xi = x.detach().requires_grad_(True)
            e = 0.5 * (self.f(xi, t, lat) ** 2).sum()
            return torch.autograd.grad(e, xi)[0].detach()


class TorchCpuSampler:
    """p_sample_loop + ddpm_sample_fn with classifier-free guidance, injected noise (same conventions as
    oracle.ramp_oracle.SamplerOracle.ddpm)."""

    def __init__(self, net: TorchCpuScoreNet, sched: Dict[str, np.ndarray], cfg_w: float = 2.0):
        self.net, self.w = net, cfg_w
        self.s = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in sched.items()}

    @torch.no_grad()
    def ddpm(self, noise: np.ndarray, hard_conds, latent: np.ndarray, cloud: Optional[np.ndarray] = None,
             use_apf: bool = False, noise_scale: float = 0.5, apf_after: int = 20) -> np.ndarray:
        from .ramp_oracle import apf_avoidance
        s, w = self.s, self.w
        T = s["betas"].shape[0]
        z = torch.from_numpy(np.ascontiguousarray(noise, np.float32))
        B = z.shape[1]
        lat = torch.from_numpy(np.asarray(latent, np.float32))[None].repeat(2 * B, 1)
        lat[1::2] = 0

        def hard(x):
            for k, v in hard_conds.items():
                x[:, k, :] = torch.from_numpy(np.asarray(v, np.float32))
            return x

        x = hard(z[0].clone())
        chain = [x.clone()]
        for j, t in enumerate(reversed(range(T))):
            out = self.net.score(x.repeat_interleave(2, dim=0), torch.full((2 * B,), t, dtype=torch.long), lat)
            out = out.view(B, 2, *out.shape[1:])
            e = (1 + w) * out[:, 0] - w * out[:, 1]
            x0 = (s["sqrt_recip_alphas_cumprod"][t] * x - s["sqrt_recipm1_alphas_cumprod"][t] * e).clamp_(-1.0, 1.0)
            mean = s["posterior_mean_coef1"][t] * x0 + s["posterior_mean_coef2"][t] * x
            if use_apf and j > apf_after:
                mean = torch.from_numpy(apf_avoidance(mean.numpy(), cloud, 0.07, 0.1, 5))
            n = z[1 + j] if t != 0 else torch.zeros_like(x)
            x = hard(mean + torch.exp(0.5 * s["posterior_log_variance_clipped"][t]) * n * noise_scale)
            chain.append(x.clone())
        return torch.stack(chain).numpy()
