#!/usr/bin/env python3
"""Drop-in counterpart of the reference's scripts/inference/inference_static.py (StaticInference.run_single_experiment,
lines 37-180) on the MI355X-native sampler.

Same flow, same calls, same on-disk layout; only the imports change (``mpd.*`` -> ``ramp_amd.*``):

    get_model(model_class=..., model=TemporalUnetInference(**unet_configs), tensor_args=..., **diffusion_configs)   :99-105
    model.load_state_dict(torch.load(<trained_models_dir>/<model_id>/checkpoints/ema_model_current_state_dict.pth))  :107-111
    model.eval(); model.warmup(horizon, traj_normalized, obstacle_pts, batch_size, device)                           :112-121
    start, goal = ContextManager.load_context(<env_dir>/contexts, context_idx); hard_conds = get_hard_cond_custom()  :123-134
    run_inference(context, hard_conds, n_samples, horizon, return_chain=True, obstacle_pts=..., sample_fn=..., ...)  :146-157
    Metrics.compute_collision_intensity / trajectory_success_and_metrics                                              :159-169

(``torch.compile(mode='reduce-overhead')``, :114, is the reference's way of removing launch overhead; here the whole
reverse-diffusion loop is one captured hipGraph inside ``run_inference``.)

The reference ships neither checkpoints nor datasets, so ``--make-synthetic DIR`` first writes an experiment tree in the
reference's layout -- ``DIR/data/<subdir>/<env>/{obstacle_points.pt, box_centers.npy, metadata.yaml, contexts/context_000.pt}``
and ``DIR/models/<model_id>/checkpoints/ema_model_current_state_dict.pth`` with seeded random weights under the reference's
state-dict keys -- and then runs the experiment on it, exactly as it would on the authors' files.

    python examples/inference_static.py --make-synthetic /tmp/ramp_exp --n-samples 64
    python examples/inference_static.py --dataset-path /data/ramp --dataset-subdir EnvSimple2D --model-id my_run --env 3
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
from ramp_amd.metrics import Metrics  # noqa: E402
from ramp_amd.models import UNET_DIM_MULTS, TemporalUnetInference  # noqa: E402
from ramp_amd.sample_functions import ddpm_sample_fn  # noqa: E402
from ramp_amd.spec import make_unet_spec  # noqa: E402
from ramp_amd.trainer import get_model  # noqa: E402


class StaticConfig:
    """The fields of the reference's config/base_config.py StaticConfig that the script reads."""
    device = "cuda"
    dataset_path = ""
    dataset_subdir = "EnvSimple2D-RobotPointMass"
    trained_models_dir = ""
    model_id = "synthetic"
    use_ema = True
    diffusion_model_class = "StaticGaussianDiffusionModel"
    variance_schedule = "exponential"
    n_diffusion_steps = 100
    predict_epsilon = True
    compose = False
    use_apf = False
    unet_input_dim = 32
    unet_dim_mults_option = 1
    include_velocity = True
    n_samples = 20
    n_support_points = 48
    state_dim = 4
    n_guide_steps = 1
    start_guide_steps_fraction = 0.07
    n_diffusion_steps_without_noise = 5
    sampler = None           # None = the reference's hard-coded default (DDIM-5); 'ddpm' for the full chain


def make_synthetic_experiment(root: str, cfg: StaticConfig, n_obstacles: int = 6, n_points: int = 64, seed: int = 42) -> None:
    """An experiment tree in the reference's on-disk layout (mpd/datasets/trajectories.py:316-347,
    scripts/inference/core/utils.py:28-75, inference_static.py:107-111) with synthetic content."""
    import yaml
    from ramp_amd.models import StaticGaussianDiffusionModel
    env_dir = os.path.join(root, "data", cfg.dataset_subdir, "0")
    os.makedirs(env_dir, exist_ok=True)
    boxes = synth.make_boxes(n_obstacles, 2, seed=seed)
    torch.save(torch.from_numpy(synth.make_cloud(n_obstacles, n_points, 2, seed=seed)), os.path.join(env_dir, "obstacle_points.pt"))
    np.save(os.path.join(env_dir, "box_centers.npy"), boxes.astype(np.float32))
    with open(os.path.join(env_dir, "metadata.yaml"), "w") as fh:
        yaml.safe_dump({"box_sizes": [[0.26, 0.26]] * n_obstacles}, fh)
    compat.ContextManager.save_context(torch.tensor([-0.8, -0.8]), torch.tensor([0.8, 0.8]), env_dir, cfg.dataset_subdir, 0)
    sp = make_unet_spec(cfg.state_dim, cfg.n_support_points)
    dm = StaticGaussianDiffusionModel(model=TemporalUnetInference(n_support_points=cfg.n_support_points, state_dim=cfg.state_dim),
                                      variance_schedule=cfg.variance_schedule, n_diffusion_steps=cfg.n_diffusion_steps,
                                      predict_epsilon=True)
    full = dm.state_dict()                                     # the 12 schedule buffers + scene encoder defaults
    for k, v in synth.make_unet_state_dict(sp, seed=0).items():
        full["model." + k] = torch.from_numpy(np.asarray(v))
    ck = os.path.join(root, "models", cfg.model_id, "checkpoints")
    os.makedirs(ck, exist_ok=True)
    torch.save(full, os.path.join(ck, "ema_model_current_state_dict.pth"))


class StaticInference:
    def __init__(self, config: StaticConfig):
        self.config = config
        self.device = config.device
        self.tensor_args = {'device': config.device, 'dtype': torch.float32}
        self.metrics_calculator = Metrics()
        self.context_manager = compat.ContextManager()
        self.model = None

    def run_single_experiment(self, current_dir: int, context_idx: int):
        cfg = self.config
        torch.cuda.set_device(0)
        env_dir = os.path.join(cfg.dataset_path, cfg.dataset_subdir, str(current_dir))
        data = compat.load_environment_dir(env_dir)            # CPU tensors, as the reference's dataset hands them over
        obstacle_pts = data['obstacle_points']
        box_centers, box_size = data['box_centers'], data['box_sizes']
        n_support_points = cfg.n_support_points
        traj_normalized = torch.zeros(n_support_points, cfg.state_dim)      # accepted and unused by the samplers
        if cfg.compose:                                        # inference_static.py:64-70
            first_batch, remaining = obstacle_pts[:6], obstacle_pts[6:]
            indices = torch.randperm(4)[:2]
            obstacle_pts = torch.stack([first_batch, torch.cat([remaining, remaining[indices]], dim=0)], dim=0)
        diffusion_configs = dict(variance_schedule=cfg.variance_schedule, n_diffusion_steps=cfg.n_diffusion_steps,
                                 predict_epsilon=cfg.predict_epsilon, compose=cfg.compose, use_apf=cfg.use_apf)
        if cfg.sampler is not None:
            diffusion_configs['sampler'] = cfg.sampler
        unet_configs = dict(state_dim=cfg.state_dim, n_support_points=n_support_points, unet_input_dim=cfg.unet_input_dim,
                            dim_mults=UNET_DIM_MULTS[cfg.unet_dim_mults_option])
        self.model = get_model(model_class=cfg.diffusion_model_class,
                               model=TemporalUnetInference(max_rows=3 * cfg.n_samples, **unet_configs),
                               tensor_args=self.tensor_args, **diffusion_configs, **unet_configs)
        compat.load_checkpoint(self.model, cfg.trained_models_dir, cfg.model_id, use_ema=cfg.use_ema, device="cpu")
        self.model.eval()
        for p in self.model.parameters():                      # freeze_torch_model_params
            p.requires_grad_(False)
        self.model.warmup(horizon=n_support_points, traj_normalized=traj_normalized, obstacle_pts=obstacle_pts,
                          batch_size=cfg.n_samples, device=self.device)
        start_state_pos, goal_state_pos = self.context_manager.load_context(os.path.join(env_dir, 'contexts'), context_idx,
                                                                            self.device)
        hard_conds = compat.StateGenerator.get_hard_cond_custom(torch.vstack((start_state_pos, goal_state_pos)),
                                                                horizon=n_support_points, include_velocity=cfg.include_velocity)
        context = {'dataset': None}
        t_start_guide = ceil(cfg.start_guide_steps_fraction * self.model.n_diffusion_steps)
        sample_fn_kwargs = dict(guide=None, n_guide_steps=cfg.n_guide_steps, t_start_guide=t_start_guide,
                                noise_std_extra_schedule_fn=lambda x: 0.5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        trajs_normalized_iters = self.model.run_inference(
            context, hard_conds, n_samples=cfg.n_samples, horizon=n_support_points, return_chain=True,
            traj_normalized=traj_normalized, obstacle_pts=obstacle_pts, sample_fn=ddpm_sample_fn, **sample_fn_kwargs,
            n_diffusion_steps_without_noise=cfg.n_diffusion_steps_without_noise)
        torch.cuda.synchronize(); elapsed = time.perf_counter() - t0
        trajs_final = trajs_normalized_iters[-1]
        collision_intensities = self.metrics_calculator.compute_collision_intensity(trajs_final, box_centers, box_size)
        metrics = self.metrics_calculator.trajectory_success_and_metrics(trajs_final, collision_intensities)
        metrics['total_time'] = elapsed
        metrics['n_chain_states'] = int(trajs_normalized_iters.shape[0])
        self.last_trajectories = trajs_final
        return metrics


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--make-synthetic", metavar="DIR", help="write a synthetic experiment tree (reference layout) to DIR and run on it")
    ap.add_argument("--dataset-path"); ap.add_argument("--dataset-subdir", default=StaticConfig.dataset_subdir)
    ap.add_argument("--trained-models-dir"); ap.add_argument("--model-id", default=StaticConfig.model_id)
    ap.add_argument("--env", type=int, default=0); ap.add_argument("--context", type=int, default=0)
    ap.add_argument("--n-samples", type=int, default=StaticConfig.n_samples)
    ap.add_argument("--n-diffusion-steps", type=int, default=StaticConfig.n_diffusion_steps)
    ap.add_argument("--sampler", choices=["ddim", "ddpm"], default=None)
    ap.add_argument("--use-apf", action="store_true")
    ap.add_argument("--n-steps-without-noise", type=int, default=StaticConfig.n_diffusion_steps_without_noise)
    args = ap.parse_args(argv)
    cfg = StaticConfig()
    cfg.n_samples, cfg.n_diffusion_steps, cfg.sampler, cfg.use_apf = args.n_samples, args.n_diffusion_steps, args.sampler, args.use_apf
    cfg.dataset_subdir, cfg.model_id = args.dataset_subdir, args.model_id
    cfg.n_diffusion_steps_without_noise = args.n_steps_without_noise
    if args.make_synthetic:
        make_synthetic_experiment(args.make_synthetic, cfg)
        cfg.dataset_path = os.path.join(args.make_synthetic, "data")
        cfg.trained_models_dir = os.path.join(args.make_synthetic, "models")
    else:
        if not (args.dataset_path and args.trained_models_dir):
            ap.error("--dataset-path and --trained-models-dir (or --make-synthetic DIR) are required")
        cfg.dataset_path, cfg.trained_models_dir = args.dataset_path, args.trained_models_dir
    runner = StaticInference(cfg)
    metrics = runner.run_single_experiment(args.env, args.context)
    print(json.dumps({k: v for k, v in metrics.items() if v is None or isinstance(v, (int, float))}))
    return metrics, runner


if __name__ == "__main__":
    main()
