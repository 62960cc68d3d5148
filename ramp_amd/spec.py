"""Architecture description of the RAMP score network (shapes only, no arithmetic).

One place that enumerates every parameter of ``TemporalUnetInference`` under the
reference's own state_dict key names, so that a real RAMP checkpoint loads
unchanged (reference: mpd/models/diffusion_models/UnetInference.py:93-145 for the
module tree, mpd/models/layers/layers.py:233-361 for the conv blocks,
mpd/models/layers/layers_attention_mini.py:83-202 for the transformer).

Used by the weight packer (`ramp_amd.weights`), the synthetic-weight generator
(`ramp_amd.synth`) and the tests.  Nothing here imports the reference.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

UNET_DIM_MULTS = {0: (1, 2, 4), 1: (1, 2, 4, 8)}  # UnetInference.py:13-16

ATTN_HEADS = 4
ATTN_DIM_HEAD = 64
ATTN_INNER = ATTN_HEADS * ATTN_DIM_HEAD  # 256
ATTN_DEPTH = 2                            # UnetInference.py:100 (depth_attn)
FF_INNER = ATTN_INNER * 4                 # 1024 (GEGLU projects to 2*FF_INNER)
TIME_DIM = 32                             # TimeEncoder(32, time_emb_dim=32)
TIME_HIDDEN = 128


@dataclass
class RTBSpec:
    """ResidualTemporalBlock (layers.py:327-361)."""
    name: str
    cin: int
    cout: int
    length: int

    @property
    def has_res_conv(self) -> bool:
        return self.cin != self.cout


@dataclass
class STSpec:
    """SpatialTransformer (layers_attention_mini.py:152-202)."""
    name: str
    channels: int
    length: int


@dataclass
class LevelSpec:
    rtb0: RTBSpec
    rtb1: RTBSpec
    st: STSpec
    resample: str | None      # key prefix of Downsample1d / Upsample1d conv, or None
    channels: int
    length: int


@dataclass
class UNetSpec:
    state_dim: int
    horizon: int
    unet_input_dim: int = 32
    dim_mults: Tuple[int, ...] = (1, 2, 4, 8)
    obstacle_3d: bool = False
    downs: List[LevelSpec] = field(default_factory=list)
    ups: List[LevelSpec] = field(default_factory=list)
    mid1: RTBSpec | None = None
    mid_st: STSpec | None = None
    mid2: RTBSpec | None = None

    @property
    def context_dim(self) -> int:
        return 256 if self.obstacle_3d else 320      # UnetInference.py:72

    @property
    def dims(self) -> List[int]:
        return [self.state_dim] + [self.unet_input_dim * m for m in self.dim_mults]

    def all_rtbs(self) -> List[RTBSpec]:
        out: List[RTBSpec] = []
        for lv in self.downs:
            out += [lv.rtb0, lv.rtb1]
        out += [self.mid1, self.mid2]
        for lv in self.ups:
            out += [lv.rtb0, lv.rtb1]
        return out

    def all_sts(self) -> List[STSpec]:
        return [lv.st for lv in self.downs] + [self.mid_st] + [lv.st for lv in self.ups]

    def tokens_per_row(self) -> int:
        return sum(st.length for st in self.all_sts())


def make_unet_spec(state_dim: int, horizon: int, unet_input_dim: int = 32,
                   dim_mults: Tuple[int, ...] = (1, 2, 4, 8), obstacle_3d: bool = False) -> UNetSpec:
    n_levels = len(dim_mults)
    if horizon % (1 << (n_levels - 1)) != 0:
        raise ValueError(f"horizon {horizon} must be divisible by {1 << (n_levels - 1)}")
    sp = UNetSpec(state_dim, horizon, unet_input_dim, tuple(dim_mults), obstacle_3d)
    dims = sp.dims
    in_out = list(zip(dims[:-1], dims[1:]))
    L = horizon
    for k, (ci, co) in enumerate(in_out):
        last = k == len(in_out) - 1
        sp.downs.append(LevelSpec(
            RTBSpec(f"downs.{k}.0", ci, co, L), RTBSpec(f"downs.{k}.1", co, co, L),
            STSpec(f"downs.{k}.3", co, L), None if last else f"downs.{k}.4.conv", co, L))
        if not last:
            L //= 2
    mid = dims[-1]
    sp.mid1 = RTBSpec("mid_block1", mid, mid, L)
    sp.mid_st = STSpec("mid_attention", mid, L)
    sp.mid2 = RTBSpec("mid_block2", mid, mid, L)
    for k, (ci, co) in enumerate(reversed(in_out[1:])):
        # UnetInference.py:127-140: is_last is never true here -> always Upsample1d
        sp.ups.append(LevelSpec(
            RTBSpec(f"ups.{k}.0", co * 2, ci, L), RTBSpec(f"ups.{k}.1", ci, ci, L),
            STSpec(f"ups.{k}.3", ci, L), f"ups.{k}.4.conv", ci, L))
        L *= 2
    assert L == horizon
    return sp


def _rtb_params(p: "OrderedDict[str, tuple]", r: RTBSpec) -> None:
    for j, ci in ((0, r.cin), (1, r.cout)):
        p[f"{r.name}.blocks.{j}.block.0.weight"] = (r.cout, ci, 5)
        p[f"{r.name}.blocks.{j}.block.0.bias"] = (r.cout,)
        p[f"{r.name}.blocks.{j}.block.2.weight"] = (r.cout,)
        p[f"{r.name}.blocks.{j}.block.2.bias"] = (r.cout,)
    p[f"{r.name}.cond_mlp.1.weight"] = (r.cout, TIME_DIM)
    p[f"{r.name}.cond_mlp.1.bias"] = (r.cout,)
    if r.has_res_conv:
        p[f"{r.name}.residual_conv.weight"] = (r.cout, r.cin, 1)
        p[f"{r.name}.residual_conv.bias"] = (r.cout,)


def _st_params(p: "OrderedDict[str, tuple]", s: STSpec, ctx_dim: int) -> None:
    D = ATTN_INNER
    p[f"{s.name}.norm.weight"] = (s.channels,)
    p[f"{s.name}.norm.bias"] = (s.channels,)
    p[f"{s.name}.proj_in.weight"] = (D, s.channels, 1)
    p[f"{s.name}.proj_in.bias"] = (D,)
    for b in range(ATTN_DEPTH):
        t = f"{s.name}.transformer_blocks.{b}"
        for n in ("to_q", "to_k", "to_v"):
            p[f"{t}.attn1.{n}.weight"] = (D, D)
        p[f"{t}.attn1.to_out.0.weight"] = (D, D)
        p[f"{t}.attn1.to_out.0.bias"] = (D,)
        p[f"{t}.ff.net.0.proj.weight"] = (2 * FF_INNER, D)
        p[f"{t}.ff.net.0.proj.bias"] = (2 * FF_INNER,)
        p[f"{t}.ff.net.2.weight"] = (D, FF_INNER)
        p[f"{t}.ff.net.2.bias"] = (D,)
        p[f"{t}.attn2.to_q.weight"] = (D, D)
        p[f"{t}.attn2.to_k.weight"] = (D, ctx_dim)
        p[f"{t}.attn2.to_v.weight"] = (D, ctx_dim)
        p[f"{t}.attn2.to_out.0.weight"] = (D, D)
        p[f"{t}.attn2.to_out.0.bias"] = (D,)
        for n in ("norm1", "norm2", "norm3"):
            p[f"{t}.{n}.weight"] = (D,)
            p[f"{t}.{n}.bias"] = (D,)
    p[f"{s.name}.proj_out.weight"] = (s.channels, D, 1)
    p[f"{s.name}.proj_out.bias"] = (s.channels,)


def scene_encoder_param_shapes(obstacle_3d: bool) -> "OrderedDict[str, tuple]":
    """Parameters + buffers of the scene encoder under ``scene_encoder.``.

    2-D: ObstacleEncoderSet (obstacle_encoder.py:94-123); 3-D: ObstacleEncoder
    (obstacle_encoder3d.py:55-75).  Shapes do not depend on the cloud size.
    """
    p: "OrderedDict[str, tuple]" = OrderedDict()
    pre = "scene_encoder."
    if not obstacle_3d:
        hd = 64
        p[pre + "pos_encoder.div_term"] = (hd // 2,)
        p[pre + "point_embedding.0.weight"] = (hd, 2)
        p[pre + "point_embedding.0.bias"] = (hd,)
        p[pre + "point_embedding.1.weight"] = (hd,)
        p[pre + "point_embedding.1.bias"] = (hd,)
        p[pre + "combined_encoder.0.weight"] = (hd, 3 * hd)
        p[pre + "combined_encoder.0.bias"] = (hd,)
        p[pre + "combined_encoder.1.weight"] = (hd,)
        p[pre + "combined_encoder.1.bias"] = (hd,)
        for i in range(3):
            for j in range(3):
                t = f"{pre}set_transformers.{i}.{j}"
                p[f"{t}.norm1.weight"] = (hd,)
                p[f"{t}.norm1.bias"] = (hd,)
                p[f"{t}.attn.qkv.weight"] = (3 * hd, hd)
                p[f"{t}.attn.proj.weight"] = (hd, hd)
                p[f"{t}.attn.proj.bias"] = (hd,)
                p[f"{t}.norm2.weight"] = (hd,)
                p[f"{t}.norm2.bias"] = (hd,)
                p[f"{t}.mlp.0.weight"] = (4 * hd, hd)
                p[f"{t}.mlp.0.bias"] = (4 * hd,)
                p[f"{t}.mlp.3.weight"] = (hd, 4 * hd)
                p[f"{t}.mlp.3.bias"] = (hd,)
        for i, d in enumerate((64, 96, 160)):
            p[f"{pre}poolings.{i}.0.weight"] = (d, hd)
            p[f"{pre}poolings.{i}.0.bias"] = (d,)
            p[f"{pre}poolings.{i}.2.weight"] = (d, d)
            p[f"{pre}poolings.{i}.2.bias"] = (d,)
    else:
        E = 256
        pp = pre + "point_processor."
        p[pp + "conv1.weight"] = (64, 3, 1)
        p[pp + "conv1.bias"] = (64,)
        p[pp + "conv2.weight"] = (E, 64, 1)
        p[pp + "conv2.bias"] = (E,)
        for n, c in (("bn1", 64), ("bn2", E)):
            p[pp + f"{n}.weight"] = (c,)
            p[pp + f"{n}.bias"] = (c,)
            p[pp + f"{n}.running_mean"] = (c,)
            p[pp + f"{n}.running_var"] = (c,)
            p[pp + f"{n}.num_batches_tracked"] = ()
        for i in range(2):
            t = f"{pre}set_transformer_blocks.{i}"
            p[f"{t}.mha.in_proj_weight"] = (3 * E, E)
            p[f"{t}.mha.in_proj_bias"] = (3 * E,)
            p[f"{t}.mha.out_proj.weight"] = (E, E)
            p[f"{t}.mha.out_proj.bias"] = (E,)
            p[f"{t}.ffn.0.weight"] = (2 * E, E)
            p[f"{t}.ffn.0.bias"] = (2 * E,)
            p[f"{t}.ffn.3.weight"] = (E, 2 * E)
            p[f"{t}.ffn.3.bias"] = (E,)
            p[f"{t}.norm1.weight"] = (E,)
            p[f"{t}.norm1.bias"] = (E,)
            p[f"{t}.norm2.weight"] = (E,)
            p[f"{t}.norm2.bias"] = (E,)
        p[pre + "output_proj.weight"] = (E, E)
        p[pre + "output_proj.bias"] = (E,)
        p[pre + "global_pooling.0.weight"] = (E, E)
        p[pre + "global_pooling.0.bias"] = (E,)
        p[pre + "global_pooling.2.weight"] = (E, E)
        p[pre + "global_pooling.2.bias"] = (E,)
    return p


def unet_param_shapes(sp: UNetSpec, with_scene_encoder: bool = True) -> "OrderedDict[str, tuple]":
    """state_dict keys -> shapes of ``TemporalUnetInference`` (no ``model.`` prefix)."""
    p: "OrderedDict[str, tuple]" = OrderedDict()
    if with_scene_encoder:
        p.update(scene_encoder_param_shapes(sp.obstacle_3d))
    p["time_mlp.encoder.1.weight"] = (TIME_HIDDEN, TIME_DIM)
    p["time_mlp.encoder.1.bias"] = (TIME_HIDDEN,)
    p["time_mlp.encoder.3.weight"] = (TIME_DIM, TIME_HIDDEN)
    p["time_mlp.encoder.3.bias"] = (TIME_DIM,)
    for lv in sp.downs:
        _rtb_params(p, lv.rtb0)
        _rtb_params(p, lv.rtb1)
        _st_params(p, lv.st, sp.context_dim)
        if lv.resample:
            p[lv.resample + ".weight"] = (lv.channels, lv.channels, 3)
            p[lv.resample + ".bias"] = (lv.channels,)
    _rtb_params(p, sp.mid1)
    _st_params(p, sp.mid_st, sp.context_dim)
    _rtb_params(p, sp.mid2)
    for lv in sp.ups:
        _rtb_params(p, lv.rtb0)
        _rtb_params(p, lv.rtb1)
        _st_params(p, lv.st, sp.context_dim)
        p[lv.resample + ".weight"] = (lv.channels, lv.channels, 4)   # ConvTranspose1d (Cin,Cout,4)
        p[lv.resample + ".bias"] = (lv.channels,)
    c0 = sp.unet_input_dim
    p["final_conv.0.block.0.weight"] = (c0, c0, 5)
    p["final_conv.0.block.0.bias"] = (c0,)
    p["final_conv.0.block.2.weight"] = (c0,)
    p["final_conv.0.block.2.bias"] = (c0,)
    p["final_conv.1.weight"] = (sp.state_dim, c0, 1)
    p["final_conv.1.bias"] = (sp.state_dim,)
    return p


SCHEDULE_BUFFERS = (  # diffusion_model_static.py:62-84
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2",
)
