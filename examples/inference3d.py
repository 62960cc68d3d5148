#!/usr/bin/env python3
"""Drop-in counterpart of the reference's scripts/inference/inference3d.py (main, lines 19-156) on the MI355X-native
sampler: Maze3D, GaussianDiffusionModel3d (classifier-free guidance w = 5.75), 3-D scene encoder.

Same flow, same calls (``mpd.*`` -> ``ramp_amd.*``):

    data = dataset[traj_id]: obstacle_points (n_obstacles, n_points, 3), box / sphere centres and sizes           :38-47
    get_model(model_class='GaussianDiffusionModel3d', model=TemporalUnetInference(..., obstacle_3d=True), ...)     :91-113
    load_state_dict(torch.load(<models>/<model_id>/checkpoints/ema_model_current_state_dict.pth))                  :115-118
    start = (-0.8, -0.25, -0.8), goal = (0.8, -0.4, 0.9); hard_conds = dataset.get_hard_conditions(..., normalize) :124-131
    run_inference(context, hard_conds, n_samples, horizon, return_chain=True, obstacle_pts=..., sample_fn=...)     :141-151
    trajs = dataset.unnormalize_trajectories(chain)                                                                :154

(The reference's 3-D wrapper indexes output rows 0 / 1 and is only valid for n_samples = 1; here every
sample gets its own cond / uncond pair, so ``--n-samples N`` equals N independent reference runs.)  Plotting is left out.
``--make-synthetic DIR`` writes an experiment tree in the reference's layout (mpd/datasets/trajectories.py:316-351) first:
``DIR/data/EnvSmall3D/0/{obstacle_points.pt, box_centers.npy, sphere_centers.npy, metadata.yaml}`` and the checkpoint.

    python examples/inference3d.py --make-synthetic /tmp/ramp_exp3d --n-samples 8
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from math import ceil

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ramp_amd import compat, synth  # noqa: E402
from ramp_amd.models import UNET_DIM_MULTS, TemporalUnetInference  # noqa: E402
from ramp_amd.sample_functions import ddpm_sample_fn  # noqa: E402
from ramp_amd.spec import make_unet_spec  # noqa: E402
from ramp_amd.trainer import get_model  # noqa: E402


class Config3d:
    """The fields of the reference's config/base_config.py Config3d that the script reads."""
    device = "cuda"
    seed = 0
    dataset_subdir = "EnvSmall3D"
    trained_models_dir = ""
    dataset_path = ""
    model_id = "synthetic3d"
    use_ema = True
    diffusion_model_class = "GaussianDiffusionModel3d"
    variance_schedule = "exponential"
    n_diffusion_steps = 25
    predict_epsilon = True
    compose = False
    unet_input_dim = 32
    unet_dim_mults_option = 1
    include_velocity = True
    n_samples = 1
    n_support_points = 48
    state_dim = 6
    n_guide_steps = 1
    start_guide_steps_fraction = 0.07
    n_diffusion_steps_without_noise = 0
    trajectory_duration = 5.0


def make_synthetic_experiment(root: str, cfg: Config3d, cloud=None, n_obstacles: int = 20, n_points: int = 200, seed: int = 43) -> None:
    import yaml
    from ramp_amd.models import GaussianDiffusionModel3d
    env_dir = os.path.join(root, "data", cfg.dataset_subdir, "0")
    os.makedirs(env_dir, exist_ok=True)
    cloud = synth.make_cloud(n_obstacles, n_points, 3, seed=seed) if cloud is None else np.asarray(cloud, np.float32)
    n_obstacles = cloud.shape[0]
    torch.save(torch.from_numpy(cloud), os.path.join(env_dir, "obstacle_points.pt"))
    centres = cloud.mean(axis=1).astype(np.float32)
    np.save(os.path.join(env_dir, "box_centers.npy"), centres[: n_obstacles // 2])
    np.save(os.path.join(env_dir, "sphere_centers.npy"), centres[n_obstacles // 2:])
    with open(os.path.join(env_dir, "metadata.yaml"), "w") as fh:
        yaml.safe_dump({"box_sizes": [[0.26, 0.26, 0.26]] * (n_obstacles // 2), "sphere_radii": [0.1] * (n_obstacles - n_obstacles // 2),
                        "limits": [[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]}, fh)
    sp = make_unet_spec(cfg.state_dim, cfg.n_support_points, obstacle_3d=True)
    dm = GaussianDiffusionModel3d(model=TemporalUnetInference(n_support_points=cfg.n_support_points, state_dim=cfg.state_dim, obstacle_3d=True),
                                  variance_schedule=cfg.variance_schedule, n_diffusion_steps=cfg.n_diffusion_steps, predict_epsilon=True)
    full = dm.state_dict()
    for k, v in synth.make_unet_state_dict(sp, seed=0).items():
        full["model." + k] = torch.from_numpy(np.asarray(v))
    ck = os.path.join(root, "models", cfg.model_id, "checkpoints")
    os.makedirs(ck, exist_ok=True)
    torch.save(full, os.path.join(ck, "ema_model_current_state_dict.pth"))


class Dataset3dView:
    """What the script uses of TrajectoryDataset3d (mpd/datasets/trajectories.py): the item's tensors, the limits
    normaliser (normalization.py:144-167) behind get_hard_conditions / unnormalize_trajectories, state_dim, n_support_points."""

    def __init__(self, env_dir: str, cfg: Config3d):
        import yaml
        self.state_dim, self.n_support_points = cfg.state_dim, cfg.n_support_points
        self.item = {"obstacle_points": torch.load(os.path.join(env_dir, "obstacle_points.pt"), map_location="cpu"),
                     "box_centers": np.load(os.path.join(env_dir, "box_centers.npy")),
                     "sphere_centers": np.load(os.path.join(env_dir, "sphere_centers.npy")),
                     "traj_normalized": torch.zeros(cfg.n_support_points, cfg.state_dim)}
        with open(os.path.join(env_dir, "metadata.yaml")) as fh:
            meta = yaml.safe_load(fh)
        self.item["box_sizes"] = torch.tensor(meta["box_sizes"]); self.item["sphere_radii"] = torch.tensor(meta["sphere_radii"])
        lo, hi = meta["limits"]
        z = [0.0] * 3 if cfg.include_velocity else []
        vmax = [1.0] * 3 if cfg.include_velocity else []
        self.normalizer = compat.LimitsNormalizer(lo + [-v for v in vmax], hi + vmax)     # positions | velocities

    def get_hard_conditions(self, pos: torch.Tensor, normalize: bool = True):
        start, goal = pos[0], pos[-1]
        start = torch.cat((start, torch.zeros_like(start))); goal = torch.cat((goal, torch.zeros_like(goal)))
        if normalize:
            self.normalizer.to(start.device)
            start, goal = self.normalizer.normalize(start), self.normalizer.normalize(goal)
        return {0: start, self.n_support_points - 1: goal}

    def unnormalize_trajectories(self, x: torch.Tensor):
        self.normalizer.to(x.device)
        return self.normalizer.unnormalize(x)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--make-synthetic", metavar="DIR"); ap.add_argument("--dataset-path"); ap.add_argument("--trained-models-dir")
    ap.add_argument("--model-id", default=Config3d.model_id); ap.add_argument("--env", type=int, default=0)
    ap.add_argument("--n-samples", type=int, default=Config3d.n_samples)
    ap.add_argument("--n-diffusion-steps", type=int, default=Config3d.n_diffusion_steps)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    cfg = Config3d(); cfg.n_samples, cfg.n_diffusion_steps, cfg.model_id, cfg.seed = args.n_samples, args.n_diffusion_steps, args.model_id, args.seed
    if args.make_synthetic:
        make_synthetic_experiment(args.make_synthetic, cfg, cloud=getattr(main, "synthetic_cloud", None))
        cfg.dataset_path = os.path.join(args.make_synthetic, "data"); cfg.trained_models_dir = os.path.join(args.make_synthetic, "models")
    else:
        if not (args.dataset_path and args.trained_models_dir):
            ap.error("--dataset-path and --trained-models-dir (or --make-synthetic DIR) are required")
        cfg.dataset_path, cfg.trained_models_dir = args.dataset_path, args.trained_models_dir
    tensor_args = {'device': cfg.device, 'dtype': torch.float32}
    dataset = Dataset3dView(os.path.join(cfg.dataset_path, cfg.dataset_subdir, str(args.env)), cfg)
    data = dataset.item
    obstacle_pts, traj_normalized = data["obstacle_points"], data["traj_normalized"]
    n_support_points = dataset.n_support_points
    diffusion_configs = dict(variance_schedule=cfg.variance_schedule, n_diffusion_steps=cfg.n_diffusion_steps,
                             predict_epsilon=cfg.predict_epsilon, training=False, compose=cfg.compose)
    unet_configs = dict(state_dim=dataset.state_dim, n_support_points=n_support_points, unet_input_dim=cfg.unet_input_dim,
                        dim_mults=UNET_DIM_MULTS[cfg.unet_dim_mults_option], obstacle_3d=True)
    model = get_model(model_class=cfg.diffusion_model_class, model=TemporalUnetInference(max_rows=2 * cfg.n_samples, **unet_configs),
                      tensor_args=tensor_args, **diffusion_configs, **unet_configs)
    compat.load_checkpoint(model, cfg.trained_models_dir, cfg.model_id, use_ema=cfg.use_ema, device="cpu")
    model.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    start_state_pos = torch.tensor([-0.8, -0.25, -0.8]).to(cfg.device)        # inference3d.py:124-125
    goal_state_pos = torch.tensor([0.8, -0.4, 0.9]).to(cfg.device)
    hard_conds = dataset.get_hard_conditions(torch.vstack((start_state_pos, goal_state_pos)), normalize=True)
    context = {'dataset': dataset}
    t_start_guide = ceil(cfg.start_guide_steps_fraction * model.n_diffusion_steps)
    sample_fn_kwargs = dict(guide=None, n_guide_steps=cfg.n_guide_steps, t_start_guide=t_start_guide, noise_std_extra_schedule_fn=lambda x: 0.5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trajs_normalized_iters = model.run_inference(context, hard_conds, n_samples=cfg.n_samples, horizon=n_support_points, return_chain=True,
                                                 traj_normalized=traj_normalized, obstacle_pts=obstacle_pts, sample_fn=ddpm_sample_fn,
                                                 **sample_fn_kwargs, n_diffusion_steps_without_noise=cfg.n_diffusion_steps_without_noise)
    torch.cuda.synchronize(); elapsed = time.perf_counter() - t0
    trajs_iters = dataset.unnormalize_trajectories(trajs_normalized_iters)
    pos_final = trajs_iters[-1][..., :3]                                       # robot.get_position
    out = {"n_chain_states": int(trajs_normalized_iters.shape[0]), "n_samples": int(pos_final.shape[0]), "total_time": elapsed,
           "start_error": float((pos_final[:, 0] - start_state_pos).abs().max()), "goal_error": float((pos_final[:, -1] - goal_state_pos).abs().max())}
    print(json.dumps(out))
    return out, trajs_normalized_iters


if __name__ == "__main__":
    main()
