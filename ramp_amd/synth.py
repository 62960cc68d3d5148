"""Deterministic synthetic inputs: weights, obstacle clouds, hard conditions, noise.

The reference ships no checkpoints and no datasets, so every
test / bench input is generated here from integer seeds with numpy's PCG64, which is
bit-reproducible across machines.  The golden generator (oracle/make_goldens.py)
loads exactly these arrays into the imported reference, so fixtures and the GPU box
agree on inputs without the reference travelling.

Cloud distribution follows the reference generator
(deps/torch_robotics/torch_robotics/environments/env_simple2dquant.py:63-96):
per box, 1/2..2/3 of the points uniform on the perimeter, the rest uniform inside.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

from .spec import UNetSpec, unet_param_shapes


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))


def make_unet_state_dict(sp: UNetSpec, seed: int = 0, with_scene_encoder: bool = True) -> "OrderedDict[str, np.ndarray]":
    """Seeded random weights under the reference key names.

    Matrices/conv kernels: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (PyTorch-default-like).
    ``proj_out`` is NOT zero (reference zero-initialises it,
    layers_attention_mini.py:185, which would blind tests to 92 % of the FLOPs —
    SURVEY.md Appendix C, Q8).  Norm scales are 1 + 0.1 N(0,1), biases 0.1 U(-1,1).
    """
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in unet_param_shapes(sp, with_scene_encoder).items():
        g = _rng(seed, key)
        if key.endswith("num_batches_tracked"):
            sd[key] = np.array(100, dtype=np.int64)
        elif key.endswith("div_term"):
            d_model = 2 * shape[0]
            sd[key] = np.exp(np.arange(0, d_model, 2, dtype=np.float32)
                             * np.float32(-(np.log(10000.0) / d_model))).astype(np.float32)
        elif key.endswith("running_var"):
            sd[key] = (0.5 + g.random(shape)).astype(np.float32)
        elif key.endswith("running_mean"):
            sd[key] = (0.1 * g.standard_normal(shape)).astype(np.float32)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            if "ups." in key and key.endswith("4.conv.weight"):
                fan_in = shape[0] * shape[2] // 2   # ConvTranspose1d: (Cin, Cout, k), 2 taps hit each output
            b = 1.0 / np.sqrt(fan_in)
            sd[key] = g.uniform(-b, b, size=shape).astype(np.float32)
        elif key.endswith(".weight"):  # norm scale
            sd[key] = (1.0 + 0.1 * g.standard_normal(shape)).astype(np.float32)
        else:  # bias
            sd[key] = (0.1 * g.uniform(-1.0, 1.0, size=shape)).astype(np.float32)
    return sd


def add_outlier_channels(sd, seed: int = 0, n_channels: int = 8, factor: float = 512.0):
    """A copy of ``sd`` with trained-transformer statistics: in every transformer block, ``n_channels`` seeded OUTPUT channels
    (rows) of ``attn1.to_out.0.weight`` and of ``ff.net.2.weight`` are scaled by ``factor`` (2^9) -- a few residual-stream
    channels two to three orders of magnitude above the rest, which is where one power-of-two scale per tensor with a
    2^8 full-precision window is most exposed.  Biases and every other tensor are untouched."""
    out = OrderedDict((k, np.array(v, copy=True)) for k, v in sd.items())
    for key in out:
        if key.endswith("attn1.to_out.0.weight") or key.endswith("ff.net.2.weight"):
            rows = _rng(seed, "outlier/" + key).choice(out[key].shape[0], size=n_channels, replace=False)
            out[key][np.sort(rows)] *= np.float32(factor)
    return out


def make_boxes(n_boxes: int, dim: int = 2, seed: int = 42, box_size: float = 0.26,
               start=None, goal=None, extent: float = 0.7) -> np.ndarray:
    """Box centres uniform in [-extent, extent]^dim, rejecting overlap with start/goal."""
    g = np.random.Generator(np.random.PCG64([seed, 7]))
    start = np.full(dim, -0.8) if start is None else np.asarray(start, dtype=np.float64)[:dim]
    goal = np.full(dim, 0.8) if goal is None else np.asarray(goal, dtype=np.float64)[:dim]
    centres = []
    while len(centres) < n_boxes:
        c = g.uniform(-extent, extent, size=dim)
        if np.all(np.abs(c - start) < box_size) or np.all(np.abs(c - goal) < box_size):
            continue
        centres.append(c)
    return np.asarray(centres)


def make_cloud(n_obstacles: int, n_points: int, dim: int = 2, seed: int = 42,
               box_size: float = 0.26, **kw) -> np.ndarray:
    """(n_obstacles, n_points, dim) float32 point cloud sampled from axis-aligned boxes."""
    centres = make_boxes(n_obstacles, dim, seed, box_size, **kw)
    g = np.random.Generator(np.random.PCG64([seed, 11]))
    half = box_size / 2.0
    out = np.zeros((n_obstacles, n_points, dim), dtype=np.float64)
    for o in range(n_obstacles):
        n_per = int(g.integers(n_points // 2, (2 * n_points) // 3 + 1))
        pts = g.uniform(-half, half, size=(n_points, dim))
        # snap the first n_per points onto a random face (perimeter / surface)
        ax = g.integers(0, dim, size=n_per)
        sign = g.integers(0, 2, size=n_per) * 2 - 1
        pts[np.arange(n_per), ax] = sign * half
        out[o] = centres[o] + pts
    return out.astype(np.float32)


def default_hard_conds(state_dim: int, horizon: int) -> Dict[int, np.ndarray]:
    """start / goal states in normalised coordinates (inference3d.py:124-125)."""
    if state_dim == 4:
        return {0: np.array([-0.8, -0.8, 0, 0], np.float32), horizon - 1: np.array([0.8, 0.8, 0, 0], np.float32)}
    if state_dim == 6:
        return {0: np.array([-0.8, -0.25, -0.8, 0, 0, 0], np.float32),
                horizon - 1: np.array([0.8, -0.4, 0.9, 0, 0, 0], np.float32)}
    raise ValueError(state_dim)


def make_noise(shape: Tuple[int, ...], seed: int = 1234) -> np.ndarray:
    g = np.random.Generator(np.random.PCG64([seed, 13]))
    return g.standard_normal(shape, dtype=np.float32)
