"""Mirror of scripts/inference/core/metrics.py (Metrics / DynamicMetrics) on the HIP kernels: the per-trajectory
reductions and the O(B^2 H) pairwise waypoint variance run on the device (ramp_traj_metrics / ramp_waypoint_variance);
the selection logic around them is the reference's."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np
import torch

from . import _lib


def _dev_traj(trajs: torch.Tensor) -> torch.Tensor:
    if trajs.device.type != "cuda":
        raise _lib.RampHipError("metrics: tensors must live on a HIP device (no CPU path)")
    assert trajs.ndim == 3
    return trajs.detach().to(torch.float32).contiguous()


def _per_traj(trajs, centers=None, sizes=None):
    t = _dev_traj(trajs)
    B, H, S = t.shape
    out = torch.empty(3, B, device=t.device)
    nb = 0 if centers is None else centers.shape[0]
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().ramp_traj_metrics(_lib.ptr(t), B, H, S, _lib.ptr(centers), _lib.ptr(sizes), nb,
                                                 out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                                 _lib.current_stream()), "ramp_traj_metrics")
    return out


class Metrics:
    """metrics.py:5-126."""

    @staticmethod
    def compute_variance_waypoints(trajs, eps=1e-8):
        t = _dev_traj(trajs)
        B, H, S = t.shape
        if B < 2:
            return torch.tensor(float("nan"), device=t.device)
        scratch = torch.empty(2 * H * ((B + 255) // 256), dtype=torch.float64, device=t.device)
        out = torch.empty(1, dtype=torch.float64, device=t.device)
        with torch.cuda.device(t.device):
            _lib.check(_lib.load().ramp_waypoint_variance(_lib.ptr(t), B, H, S, _lib.ptr(scratch), _lib.ptr(out),
                                                          _lib.current_stream()), "ramp_waypoint_variance")
        return out[0].to(torch.float32)

    @staticmethod
    def compute_smoothness(trajs: torch.Tensor, trajs_vel: Optional[torch.Tensor] = None) -> torch.Tensor:
        if trajs_vel is not None:                      # velocities given separately: pad two position columns
            assert trajs_vel.ndim == 3
            trajs = torch.cat([torch.zeros_like(trajs_vel[..., :2]), trajs_vel], dim=-1)
        return _per_traj(trajs)[2]

    @staticmethod
    def compute_path_length(trajectories: torch.Tensor) -> torch.Tensor:
        assert trajectories.ndim == 3
        if len(trajectories) == 0:
            return torch.tensor(0.0, device=trajectories.device)
        return _per_traj(trajectories)[1]

    @staticmethod
    def compute_collision_intensity(trajs: torch.Tensor, box_centers, box_sizes) -> torch.Tensor:
        dev = trajs.device
        c = torch.as_tensor(box_centers, dtype=torch.float32, device=dev)
        s = torch.as_tensor(box_sizes, dtype=torch.float32, device=dev)
        if s.dim() == 1:
            s = s.unsqueeze(-1).repeat(1, 2)
        return _per_traj(trajs, c.reshape(-1, 2)[:, :2].contiguous(), s.reshape(-1, 2).contiguous())[0]

    def trajectory_success_and_metrics(self, trajs_final: torch.Tensor, collision_intensities: torch.Tensor,
                                       threshold: float = 0.01) -> Dict[str, Any]:
        ok = collision_intensities <= threshold
        free = trajs_final[torch.where(ok)[0]]
        n_free = len(free)
        m = {'success': 1 if bool(torch.any(ok)) else 0,
             'collision_intensity': collision_intensities.mean().item() * 100,
             'path_length': None, 'path_length_std': None, 'waypoint_variance': None,
             'free_trajectories': free, 'n_free_trajectories': n_free}
        if n_free > 0:
            pl = self.compute_path_length(free)
            m['path_length'] = pl.mean().item()
            m['path_length_std'] = pl.std().item()
            if n_free == 1:
                m['waypoint_variance'] = 0.0
            else:
                v = float(self.compute_variance_waypoints(free))
                m['waypoint_variance'] = v if not np.isnan(v) else None
        return m


class DynamicMetrics(Metrics):
    """metrics.py:128-170: host-side bookkeeping over the executed states of one pursuit-evasion episode."""

    def calculate_single_episode_metrics(self, chain_start: List, chain_obs: List, start_state_pos, goal_state_pos,
                                         goal_safe_threshold: float, static_collision: bool,
                                         pursuer_radius: float) -> Dict[str, Any]:
        goal = goal_state_pos.cpu().numpy() if torch.is_tensor(goal_state_pos) else goal_state_pos
        thr = pursuer_radius + 0.02
        capture = False
        for i in range(len(chain_obs)):
            if i + 2 >= len(chain_start):
                break
            if np.linalg.norm(chain_start[i + 2] - chain_obs[i]) <= thr:
                capture = True
                break
        captured = static_collision or capture
        reached = (np.linalg.norm(chain_start[-1] - goal) <= goal_safe_threshold) and not captured
        plen = 0
        for i in range(len(chain_start) - 1):
            plen += np.linalg.norm(chain_start[i + 1] - chain_start[i])
        return {'static_collision': static_collision, 'pursuer_capture': capture, 'captured': captured,
                'goal_reached': reached, 'path_length': plen if not captured else None,
                'score': 0.5 * float(reached) + 0.5 * float(not captured)}
