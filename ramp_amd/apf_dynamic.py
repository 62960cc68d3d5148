"""Mirror of mpd/models/diffusion_models/APFhelper_dynamic.py (pursuit-evasion APF) on the HIP kernel.

``ObstacleField`` keeps the static box cloud and the moving pursuer cloud as float64 device arrays (the reference
keeps them as numpy float64 behind two scipy cKDTrees); ``avoidance`` runs the per-trajectory static / pursuer
pass for one trajectory (H,S) like the reference, or for a whole batch (B,H,S) in one launch.
The reference re-samples the clouds with unseeded ``np.random``; here the clouds can also
be supplied explicitly, which is what the parity fixtures do.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch

from . import _lib


def generate_sphere_points(center, radius, num_points, surface_ratio=0.9):
    """Golden-angle ring of ``surface_ratio * num_points`` points plus uniform interior points
    (APFhelper_dynamic.py:18-39)."""
    n_surf = int(num_points * surface_ratio)
    n_in = num_points - n_surf
    ang = np.pi * (3 - np.sqrt(5)) * np.arange(n_surf)
    xs, ys = radius * np.cos(ang) + center[0], radius * np.sin(ang) + center[1]
    if n_in > 0:
        r = radius * np.sqrt(np.random.uniform(0, 1, n_in))
        th = np.random.uniform(0, 2 * np.pi, n_in)
        xs = np.concatenate((xs, r * np.cos(th) + center[0]))
        ys = np.concatenate((ys, r * np.sin(th) + center[1]))
    return np.column_stack((xs, ys))


def generate_box_points(center, size, num_points):
    """2/3..all of the points uniform on the box perimeter, the rest uniform inside (APFhelper_dynamic.py:41-68)."""
    (cx, cy), (w, h) = center, size
    left, right, top, bottom = cx - w / 2, cx + w / 2, cy + h / 2, cy - h / 2
    n_b = np.random.randint(2 * num_points // 3, num_points + 1)
    n_i = num_points - n_b
    corners = np.array([[left, top], [right, top], [right, bottom], [left, bottom]])
    lens = np.array([w, h, w, h]).repeat(2)
    pos = np.random.rand(n_b) * lens.sum()
    cum = np.cumsum(lens)
    e = np.searchsorted(cum, pos)
    t = (pos - np.concatenate(([0], cum[:-1]))[e]) / lens[e]
    a, b = corners[e % 4], corners[(e + 1) % 4]
    boundary = a + t[:, None] * (b - a)
    inside = np.random.rand(n_i, 2)
    inside[:, 0] = inside[:, 0] * w + left
    inside[:, 1] = inside[:, 1] * h + bottom
    return np.concatenate([boundary, inside], axis=0)


class ObstacleField:
    def __init__(self, static_obstacle_centers=None, static_obstacle_sizes=None,
                 dynamic_obstacle_fn: Optional[Callable] = None, points_per_obstacle=32, distance_threshold=0.1,
                 distance_threshold_pred=0.2, static_points=None, device="cuda"):
        self.dynamic_obstacle_fn = dynamic_obstacle_fn
        self.points_per_obstacle = points_per_obstacle
        self.distance_threshold = float(distance_threshold)
        self.distance_threshold_pred = float(distance_threshold_pred)
        self.device = torch.device(device)
        if static_points is None:
            static_points = np.vstack([generate_box_points(c, s, points_per_obstacle)
                                       for c, s in zip(static_obstacle_centers, static_obstacle_sizes)])
        self.static_obstacle_points = np.asarray(static_points, np.float64)
        self._static_dev = torch.from_numpy(self.static_obstacle_points).to(self.device).contiguous()
        self.dynamic_points = None
        self._dynamic_dev = None
        self.dynamic_center = None
        self.last_update_time = None

    def set_dynamic_points(self, points):
        self.dynamic_points = np.asarray(points, np.float64)
        self._dynamic_dev = torch.from_numpy(self.dynamic_points).to(self.device).contiguous()

    def update_dynamic(self, t, start_pos, replan_guide=False, best_idx=None):
        if self.last_update_time != t:
            center, radius = self.dynamic_obstacle_fn(t, start_pos, replan_guide, best_idx)
            self.dynamic_center = center
            self.set_dynamic_points(generate_sphere_points(center, radius, self.points_per_obstacle))
            self.last_update_time = t


def avoidance(trajectory, obstacle_field: ObstacleField, is_dynamic=False, avoidance_window=5, avoidance_strength=0.1,
              avoidance_strength_pred=0.3, affected_states=5, goal_state=None, stepp=None, enable=None):
    """In place, like the reference.  trajectory (H,S) or (B,H,S) float32 on the HIP device."""
    if trajectory.device.type != "cuda":
        raise _lib.RampHipError("avoidance: trajectory must live on a HIP device (no CPU path)")
    tr = trajectory if trajectory.dim() == 3 else trajectory.unsqueeze(0)
    if not tr.is_contiguous() or tr.dtype != torch.float32:
        raise ValueError("trajectory must be contiguous float32")
    B, H, S = tr.shape
    if is_dynamic:
        if obstacle_field._dynamic_dev is None:
            return trajectory
        pts, thr_q, strength = obstacle_field._dynamic_dev, obstacle_field.distance_threshold_pred, avoidance_strength_pred
        window, affected = -1, H if affected_states is None else int(affected_states)
    else:
        pts, thr_q, strength = obstacle_field._static_dev, obstacle_field.distance_threshold, avoidance_strength
        window, affected = int(avoidance_window), H
    goal = None if goal_state is None else goal_state.to(tr.device, torch.float32).contiguous()
    en = None if enable is None else enable.to(tr.device, torch.int32).contiguous()
    with torch.cuda.device(tr.device):
        _lib.check(_lib.load().ramp_apf_dynamic(_lib.ptr(tr), B, H, S, _lib.ptr(pts), pts.shape[0], thr_q,
                                                obstacle_field.distance_threshold, float(strength), window, affected,
                                                _lib.ptr(goal), _lib.ptr(en), _lib.current_stream()),
                   "ramp_apf_dynamic")
    return trajectory
