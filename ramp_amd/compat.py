"""Checkpoint / dataset / environment compatibility layer: what the reference's inference scripts
touch around the sampler, so the drop-in can run the authors' experiment directories unchanged.

  load_checkpoint        scripts/inference/inference_static.py:107-111  (state_dict .pth, reference key names)
  load_environment_dir   mpd/datasets/trajectories.py:316-347           (obstacle_points.pt, box_centers.npy, metadata.yaml)
  LimitsNormalizer       mpd/datasets/normalization.py:144-167
  StateGenerator / ContextManager / DynamicsGenerator   scripts/inference/core/utils.py:6-137
  PursuerField / make_pursuit_env   torch_robotics primitives.py:90-107 (MultiSphereFieldDynamics.update_centers) and the
                         attribute path the dynamic planner walks (context['dataset'].env.obj_fixed_list / obj_extra_list)
Host-side I/O and bookkeeping only; every tensor operation on trajectories stays on the HIP kernels.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, Tuple

import numpy as np
import torch


def load_checkpoint(model: torch.nn.Module, trained_models_dir: str, model_id: str, use_ema: bool = True,
                    device="cpu") -> torch.nn.Module:
    name = 'ema_model_current_state_dict.pth' if use_ema else 'model_current_state_dict.pth'
    sd = torch.load(os.path.join(trained_models_dir, model_id, 'checkpoints', name), map_location=device)
    model.load_state_dict(sd)
    return model.eval()


def load_environment_dir(env_dir: str, device="cpu") -> Dict[str, torch.Tensor]:
    """One experiment directory: the obstacle cloud (n_obstacles, n_points, dim), the box centres and sizes."""
    import yaml
    out = {"obstacle_points": torch.load(os.path.join(env_dir, 'obstacle_points.pt'), map_location=device),
           "box_centers": torch.from_numpy(np.load(os.path.join(env_dir, 'box_centers.npy'))).to(device, torch.float32)}
    with open(os.path.join(env_dir, 'metadata.yaml')) as fh:
        out["box_sizes"] = torch.tensor(yaml.safe_load(fh)['box_sizes'], dtype=torch.float32, device=device)
    return out


class LimitsNormalizer:
    """maps [xmin, xmax] to [-1, 1] (normalization.py:144-167)."""

    def __init__(self, mins, maxs):
        self.mins = torch.as_tensor(mins, dtype=torch.float32)
        self.maxs = torch.as_tensor(maxs, dtype=torch.float32)

    def to(self, device):
        self.mins, self.maxs = self.mins.to(device), self.maxs.to(device)
        return self

    def normalize(self, x):
        return 2 * ((x - self.mins) / (self.maxs - self.mins)) - 1

    def unnormalize(self, x, eps=1e-4):
        if x.max() > 1 + eps or x.min() < -1 - eps:
            x = torch.clip(x, -1, 1)
        return (x + 1) / 2. * (self.maxs - self.mins) + self.mins


class StateGenerator:
    @staticmethod
    def get_hard_cond_custom(traj: torch.Tensor, horizon: int, include_velocity: bool = True):
        start, goal = traj[0], traj[-1]
        if include_velocity:
            start = torch.cat((start, torch.zeros_like(start)), dim=-1)
            goal = torch.cat((goal, torch.zeros_like(goal)), dim=-1)
        return {0: start, horizon - 1: goal}


class ContextManager:
    @staticmethod
    def save_context(start_state_pos: torch.Tensor, goal_state_pos: torch.Tensor, env_dir: str, dataset_id: str,
                     context_idx: int) -> str:
        if not torch.is_tensor(start_state_pos) or not torch.is_tensor(goal_state_pos):
            raise ValueError("Start and goal positions must be torch tensors")
        if not isinstance(context_idx, int) or context_idx < 0:
            raise ValueError(f"Invalid context_idx: {context_idx}")
        d = os.path.join(env_dir, 'contexts')
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f'context_{context_idx:03d}.pt')
        torch.save({'start_pos': start_state_pos.cpu(), 'goal_pos': goal_state_pos.cpu(),
                    'metadata': {'context_idx': context_idx, 'dataset_id': dataset_id}}, path)
        return path

    @staticmethod
    def load_context(contexts_dir: str, context_idx: int, device: str = 'cpu') -> Tuple[torch.Tensor, torch.Tensor]:
        cfg = torch.load(os.path.join(contexts_dir, f'context_{context_idx:03d}.pt'), map_location=device)
        return cfg['start_pos'].to(device), cfg['goal_pos'].to(device)


class DynamicsGenerator:
    @staticmethod
    def create_pursuit_dynamics(velocity_max: float = 0.5, pursuit_strength: float = 0.8, random_strength: float = 0.2):
        """Pursuer step: unit vector toward the evader blended with a unit circular drift, scaled per axis by the
        pursuer's velocity and dt = 0.1, clipped to the [-1, 1] workspace (utils.py:85-137)."""
        velocity = np.array([[velocity_max / np.sqrt(2), velocity_max / np.sqrt(2)]])

        def dynamics_fn(t, prev_center, robot_position, velocity_input):
            dt = 0.1
            x, y = prev_center[0]
            vx, vy = velocity_input[0]
            rx, ry = robot_position[0]
            dx, dy = rx - x, ry - y
            dist = np.sqrt(dx ** 2 + dy ** 2)
            if dist > 0:
                dx, dy = dx / dist, dy / dist
            dx = pursuit_strength * dx + random_strength * np.sin(2 * np.pi * t)
            dy = pursuit_strength * dy + random_strength * np.cos(2 * np.pi * t)
            return np.array([[np.clip(x + dx * vx * dt, -1, 1), np.clip(y + dy * vy * dt, -1, 1)]])

        return dynamics_fn, velocity


class PursuerField:
    """The moving sphere(s) of the pursuit-evasion environment: .centers (n, 2), .radii (n,), update_centers(t, state)."""

    def __init__(self, centers, radii, dynamics_fn=None, velocity=None, device="cpu"):
        self.device = torch.device(device)
        self.initial_centers = centers
        self.centers = torch.as_tensor(np.asarray(centers), dtype=torch.float32, device=self.device)
        self.radii = torch.as_tensor(np.asarray(radii), dtype=torch.float32, device=self.device)
        self.velocity = None if velocity is None else torch.as_tensor(np.asarray(velocity), dtype=torch.float32,
                                                                      device=self.device)
        self.dynamics_fn = dynamics_fn

    def update_centers(self, t, current_state=None):
        if self.dynamics_fn is not None and t is not None and current_state is not None:
            new = self.dynamics_fn(t, self.centers.cpu().numpy(), current_state.detach().cpu().numpy(),
                                   self.velocity.cpu().numpy())
            self.centers = torch.as_tensor(new, dtype=torch.float32, device=self.device)


def make_pursuit_env(box_centers, box_sizes, pursuer_center, pursuer_radius=0.1, velocity_max=0.5, device="cpu"):
    """``context['dataset']`` stand-in with the attribute path the dynamic planner walks."""
    fn, vel = DynamicsGenerator.create_pursuit_dynamics(velocity_max)
    boxes = SimpleNamespace(centers=torch.as_tensor(np.asarray(box_centers), dtype=torch.float32, device=device),
                            sizes=torch.as_tensor(np.asarray(box_sizes), dtype=torch.float32, device=device))
    pursuer = PursuerField(np.asarray(pursuer_center, np.float32).reshape(1, 2), np.array([pursuer_radius], np.float32),
                           fn, vel, device=device)
    env = SimpleNamespace(obj_fixed_list=[SimpleNamespace(fields=[boxes])], obj_extra_list=[SimpleNamespace(fields=[pursuer])])
    return SimpleNamespace(env=env)
