"""Scene-encoder PARAMETER CONTAINERS.  The arithmetic runs in HIP (``ramp_encode_scene`` / csrc/scene.hip, reached through
``TemporalUnetInference.encode_scene``); these torch modules only exist so that ``state_dict`` / ``load_state_dict`` /
``.to(device)`` behave like the reference's and a real RAMP checkpoint loads unchanged.  They have no ``forward``: calling
one raises.  (A torch restatement of the two encoders, used as a CPU cross-check of the fixtures, lives with the tests:
tests/torch_scene_encoders.py.)

Same parameter names / shapes as the reference:
  2-D  ObstacleEncoderSet   mpd/models/diffusion_models/obstacle_encoder.py:94-152
  3-D  ObstacleEncoder      mpd/models/diffusion_models/obstacle_encoder3d.py:55-94
Unlike the reference's ``cache_scene_encoding`` (UnetInference.py:146-156), which encodes the same cloud once per network row
(2B identical scenes), only the distinct scenes are encoded.
"""
from __future__ import annotations

import math

import torch
from torch import nn


class _Container(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} only holds parameters: the scene encoders run in HIP "
                           "(TemporalUnetInference.encode_scene -> ramp_encode_scene); there is no torch / CPU path")


class _SetAttention(_Container):
    def __init__(self, dim: int, num_heads: int = 4):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)


class _SetBlock2d(_Container):
    def __init__(self, dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _SetAttention(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = nn.Sequential(nn.Linear(dim, dim * 4), nn.GELU(), nn.Identity(), nn.Linear(dim * 4, dim),
                                 nn.Identity())


class _PosEnc(_Container):
    def __init__(self, d_model: int):
        super().__init__()
        self.d_model = d_model
        self.register_buffer("div_term", torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model)))


class ObstacleEncoderSet(_Container):
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


class _PointProcessor(_Container):
    def __init__(self, input_dim=3, output_dim=256):
        super().__init__()
        self.conv1 = nn.Conv1d(input_dim, 64, 1)
        self.conv2 = nn.Conv1d(64, output_dim, 1)
        self.bn1 = nn.BatchNorm1d(64)
        self.bn2 = nn.BatchNorm1d(output_dim)


class _SetBlock3d(_Container):
    def __init__(self, dim=256, num_heads=4, dropout=0.1):
        super().__init__()
        self.mha = nn.MultiheadAttention(dim, num_heads, dropout=dropout)
        self.ffn = nn.Sequential(nn.Linear(dim, dim * 2), nn.SELU(), nn.Dropout(dropout), nn.Linear(dim * 2, dim))
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)


class ObstacleEncoder(_Container):
    """3-D scene encoder -> 256-d latent (eval-mode BatchNorm uses the running statistics)."""

    def __init__(self, embedding_dim=256, point_dim=3, num_layers=2):
        super().__init__()
        self.point_processor = _PointProcessor(point_dim, embedding_dim)
        self.set_transformer_blocks = nn.ModuleList([_SetBlock3d(embedding_dim) for _ in range(num_layers)])
        self.output_proj = nn.Linear(embedding_dim, embedding_dim)
        self.global_pooling = nn.Sequential(nn.Linear(embedding_dim, embedding_dim), nn.SELU(),
                                            nn.Linear(embedding_dim, embedding_dim))
        self.embedding_dim = embedding_dim
