"""Scene-encoder parameter containers (the arithmetic runs in HIP: ramp_encode_scene / csrc/scene.hip).

These torch modules exist so that ``state_dict`` / ``load_state_dict`` / ``.to(device)`` behave like the
reference's; their ``forward`` is a torch restatement used only by the CPU tests as a cross-check.

Same parameter names / shapes as the reference so checkpoints load unchanged:
  2-D  ObstacleEncoderSet   mpd/models/diffusion_models/obstacle_encoder.py:94-152
  3-D  ObstacleEncoder      mpd/models/diffusion_models/obstacle_encoder3d.py:55-94
Unlike the reference's ``cache_scene_encoding`` (UnetInference.py:146-156), which encodes the
same cloud once per network row (2B identical scenes), only the distinct scenes are encoded.
SURVEY.md §8 row a-11 ("keep in PyTorch-ROCm first, HIP later").
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn


class _SetAttention(nn.Module):
    def __init__(self, dim: int, num_heads: int = 4):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class _SetBlock2d(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _SetAttention(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = nn.Sequential(nn.Linear(dim, dim * 4), nn.GELU(), nn.Identity(), nn.Linear(dim * 4, dim),
                                 nn.Identity())

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PosEnc(nn.Module):
    def __init__(self, d_model: int):
        super().__init__()
        self.d_model = d_model
        self.register_buffer("div_term", torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model)))

    def _pe(self, v):
        out = torch.zeros(*v.shape[:-1], self.d_model, device=v.device, dtype=v.dtype)
        out[..., 0::2] = torch.sin(v[..., 0, None] * self.div_term) + torch.sin(v[..., 1, None] * self.div_term)
        out[..., 1::2] = torch.cos(v[..., 0, None] * self.div_term) + torch.cos(v[..., 1, None] * self.div_term)
        return out

    def forward(self, x):
        b, no, npnt, _ = x.shape
        centres = x.mean(dim=2)
        rel = x - centres.unsqueeze(2)
        maxd, _ = torch.max(torch.abs(rel).view(b, no, -1), dim=-1, keepdim=True)
        return self._pe(centres), self._pe(rel / (maxd.unsqueeze(-1) + 1e-8))


class ObstacleEncoderSet(nn.Module):
    """2-D scene encoder -> 64 + 96 + 160 = 320-d latent; any (num_obstacles, num_points)."""

    def __init__(self, input_dim=2, hidden_dim=64, output_dims=(64, 96, 160), num_blocks=3, **_):
        super().__init__()
        self.pos_encoder = _PosEnc(hidden_dim)
        self.point_embedding = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.GELU())
        self.combined_encoder = nn.Sequential(nn.Linear(hidden_dim * 3, hidden_dim), nn.LayerNorm(hidden_dim), nn.GELU())
        self.set_transformers = nn.ModuleList(
            [nn.Sequential(*[_SetBlock2d(hidden_dim) for _ in range(num_blocks)]) for _ in output_dims])
        self.poolings = nn.ModuleList(
            [nn.Sequential(nn.Linear(hidden_dim, d), nn.GELU(), nn.Linear(d, d)) for d in output_dims])

    def forward(self, x):
        b, no, npnt, _ = x.shape
        pe_obs, pe_rel = self.pos_encoder(x)
        emb = self.point_embedding(x.reshape(b * no * npnt, -1)).view(b, no, npnt, -1)
        comb = torch.cat([emb, pe_obs.unsqueeze(2).expand(-1, -1, npnt, -1), pe_rel], dim=-1)
        comb = self.combined_encoder(comb).view(b, no * npnt, -1)
        outs = [pool(tr(comb).mean(dim=1)) for tr, pool in zip(self.set_transformers, self.poolings)]
        return torch.cat(outs, dim=-1)


class _PointProcessor(nn.Module):
    def __init__(self, input_dim=3, output_dim=256):
        super().__init__()
        self.conv1 = nn.Conv1d(input_dim, 64, 1)
        self.conv2 = nn.Conv1d(64, output_dim, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(output_dim)

    def forward(self, x):
        x = x.transpose(2, 1)
        x = F.selu(self.bn1(self.conv1(x)))
        x = F.selu(self.bn2(self.conv2(x)))
        return torch.max(x, 2)[0]


class _SetBlock3d(nn.Module):
    def __init__(self, dim=256, num_heads=4, dropout=0.1):
        super().__init__()
        self.mha = nn.MultiheadAttention(dim, num_heads, dropout=dropout)
        self.ffn = nn.Sequential(nn.Linear(dim, dim * 2), nn.SELU(), nn.Dropout(dropout), nn.Linear(dim * 2, dim))
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)

    def forward(self, x):
        xn = self.norm1(x).transpose(0, 1)
        a, _ = self.mha(xn, xn, xn)
        x = x + a.transpose(0, 1)
        return x + self.ffn(self.norm2(x))


class ObstacleEncoder(nn.Module):
    """3-D scene encoder -> 256-d latent (eval-mode BatchNorm uses the running statistics)."""

    def __init__(self, embedding_dim=256, point_dim=3, num_layers=2):
        super().__init__()
        self.point_processor = _PointProcessor(point_dim, embedding_dim)
        self.set_transformer_blocks = nn.ModuleList([_SetBlock3d(embedding_dim) for _ in range(num_layers)])
        self.output_proj = nn.Linear(embedding_dim, embedding_dim)
        self.global_pooling = nn.Sequential(nn.Linear(embedding_dim, embedding_dim), nn.SELU(),
                                            nn.Linear(embedding_dim, embedding_dim))
        self.embedding_dim = embedding_dim

    def forward(self, pts):
        b, no, npnt, d = pts.shape
        x = self.point_processor(pts.reshape(-1, npnt, d)).view(b, no, self.embedding_dim)
        for blk in self.set_transformer_blocks:
            x = blk(x)
        feat = self.output_proj(x)
        return self.global_pooling(torch.max(feat, dim=1)[0])
