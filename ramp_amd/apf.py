"""Mirror of mpd/models/diffusion_models/APFhelper.py (static artificial potential field) on the HIP kernel.

The reference builds a scipy cKDTree on the host and round-trips every waypoint through numpy
(APFhelper.py:46-54); here the nearest obstacle point is a brute-force float64 min-reduction on the
GPU (identical result except on exact distance ties) and the window scatter is a gather.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class ObstacleField:
    def __init__(self, obstacle_pts, distance_threshold=0.1, max_leaf_size=16):
        self.distance_threshold = float(distance_threshold)
        pts = obstacle_pts if torch.is_tensor(obstacle_pts) else torch.as_tensor(obstacle_pts)
        self.obstacle_points_tensor = pts.reshape(-1, 2).to(torch.float32)
        self.static_obstacle_points = self.obstacle_points_tensor


def window_weights(avoidance_window: int) -> torch.Tensor:
    """APFhelper.py:42-44 (same torch expression, float32)."""
    return torch.exp(-0.5 * torch.square(torch.arange(-avoidance_window, avoidance_window + 1))
                     / (avoidance_window / 2) ** 2).float()


def avoidance(trajectories, obstacle_field: ObstacleField, avoidance_window=7, avoidance_strength=0.2, passes=1):
    """Returns a modified copy of ``trajectories`` (B,H,S): xy channels pushed away from the nearest cloud
    point within the threshold, spread over +-window waypoints (APFhelper.py:37-104)."""
    if trajectories.device.type != "cuda":
        raise _lib.RampHipError("avoidance: trajectories must live on a HIP device (no CPU path)")
    out = trajectories.detach().to(torch.float32).clone().contiguous()
    B, H, S = out.shape
    cloud = obstacle_field.obstacle_points_tensor.to(out.device).contiguous()
    w = window_weights(avoidance_window).contiguous()
    p = _lib.RampApfParams()
    p.cloud = _lib.ptr(cloud)
    p.n_points = cloud.shape[0]
    p.window = int(avoidance_window)
    p.window_weights_host = C.cast(w.data_ptr(), _lib.c_f32p)
    p.threshold = obstacle_field.distance_threshold
    p.strength = float(avoidance_strength)
    p.passes = int(passes)
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().ramp_apf(_lib.ptr(out), B, H, S, C.byref(p), _lib.current_stream()), "ramp_apf")
    return out
