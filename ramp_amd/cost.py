"""Mirror of mpd/models/diffusion_models/cost.py on the HIP kernel (distance reduction on the GPU,
normalisation / argmin over B scalars in torch)."""
from __future__ import annotations

import torch

from . import _lib


def _costs(trajs, obstacle_points, thr):
    if trajs.device.type != "cuda":
        raise _lib.RampHipError("trajectory costs: tensors must live on a HIP device (no CPU path)")
    t = trajs.detach().to(torch.float32).contiguous()
    B, H, S = t.shape
    cloud = obstacle_points.reshape(-1, 2).to(t.device, torch.float32).contiguous()
    mask = torch.empty(B, dtype=torch.int32, device=t.device)
    plen = torch.empty(B, device=t.device)
    smooth = torch.empty(B, device=t.device)
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().ramp_traj_costs(_lib.ptr(t), B, H, S, _lib.ptr(cloud), cloud.shape[0], float(thr),
                                               _lib.ptr(mask), _lib.ptr(plen), _lib.ptr(smooth),
                                               _lib.current_stream()), "ramp_traj_costs")
    return mask.bool(), plen, smooth


def compute_collision_with_pointcloud(trajs, obstacle_points, collision_threshold=0.0, safety_margin=0.05):
    """cost.py:25-54."""
    return _costs(trajs, obstacle_points, collision_threshold)[0]


def compute_path_length(trajs):
    """cost.py:3-7."""
    return _costs(trajs, torch.full((1, 2), 1e9, device=trajs.device), 0.0)[1]


def compute_smoothness(trajs):
    """cost.py:19-24."""
    return _costs(trajs, torch.full((1, 2), 1e9, device=trajs.device), 0.0)[2]


def compute_trajectory_costs(trajs, obstacle_points, smoothness_weight=.1, path_length_weight=.9,
                             collision_threshold=0.0, normalize=True):
    """cost.py:56-88: (best_trajectory, best_cost, total_costs, collision_free_mask, best_index)."""
    coll, plen, smooth = _costs(trajs, obstacle_points, collision_threshold)
    free = ~coll
    if not free.any():
        return None, None, None, free, None
    pl, sm = plen[free], smooth[free]
    if normalize:
        pl = (pl - pl.min()) / (pl.max() - pl.min())
        sm = (sm - sm.min()) / (sm.max() - sm.min())
    total = smoothness_weight * sm + path_length_weight * pl
    best = torch.argmin(total)
    return trajs[free][best], total[best], total, free, best
