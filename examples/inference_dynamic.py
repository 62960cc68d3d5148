#!/usr/bin/env python3
"""Drop-in counterpart of the reference's scripts/inference/inference_dynamic.py (DynamicInference.run_single_experiment,
lines 105-275) on the MI355X-native sampler: pursuit-evasion replanning with DynamicGaussianDiffusionModel.

Same flow, same calls (``mpd.*`` -> ``ramp_amd.*``):

    dataset with a static box environment + one pursuer sphere whose centre follows dynamics_fn                    :111-127
    obstacle_pts = cat(obstacle_pts[:4], obstacle_pts[randint(0, 4, (2,))])      (4 obstacles + 2 repeats)         :142
    start, goal = ContextManager.load_context(<dataset>/<subdir>/contexts/contexts, context_idx)                    :150-154
    get_model(model_class='DynamicGaussianDiffusionModel', model=TemporalUnetInference(**unet_configs), ...)       :160-178
    load_state_dict(torch.load(.../checkpoints/ema_model_current_state_dict.pth)); eval; freeze                    :181-188
    hard_conds = StateGenerator.get_hard_cond_custom(vstack(start, goal), horizon, include_velocity)               :190-194
    chain, chain_obs, chain_start = run_inference(context, hard_conds, n_samples, horizon, return_chain=True, ...) :211-222
    executed path = chain_start; Metrics.compute_collision_intensity(path, box_centers[:4], box_size[:4])          :224-252

Every replan is one captured hipGraph (ramp_replan); the environment callback (pursuer dynamics) runs on the host between
replays.  ``--make-synthetic DIR`` writes an experiment tree in the reference's layout first (static boxes, context file,
checkpoint).  Rendering is left out.

    python examples/inference_dynamic.py --make-synthetic /tmp/ramp_dyn --n-samples 64 --max-replans 6
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


class DynamicConfig:
    """The fields of the reference's config/base_config.py DynamicConfig that the script reads."""
    device = "cuda"
    dataset_path = ""
    dataset_subdir = "EnvSimple2D-RobotPointMass"
    trained_models_dir = ""
    model_id = "synthetic_dyn"
    use_ema = True
    diffusion_model_class = "DynamicGaussianDiffusionModel"
    variance_schedule = "exponential"
    n_diffusion_steps = 100
    predict_epsilon = True
    unet_input_dim = 32
    unet_dim_mults_option = 1
    include_velocity = True
    n_samples = 30
    n_support_points = 48
    state_dim = 4
    n_guide_steps = 1
    start_guide_steps_fraction = 0.07
    n_diffusion_steps_without_noise = 5
    trajectory_duration = 5.0
    pursuer_pos = (0.6, 0.55)
    pursuer_radius = 0.1
    velocity_max = 0.5
    max_replans = 60


def make_synthetic_experiment(root: str, cfg: DynamicConfig, seed: int = 42) -> None:
    import yaml
    from ramp_amd.models import DynamicGaussianDiffusionModel
    env_dir = os.path.join(root, "data", cfg.dataset_subdir, "0")
    os.makedirs(env_dir, exist_ok=True)
    boxes = synth.make_boxes(6, 2, seed=seed)
    torch.save(torch.from_numpy(synth.make_cloud(6, 64, 2, seed=seed)), os.path.join(env_dir, "obstacle_points.pt"))
    np.save(os.path.join(env_dir, "box_centers.npy"), boxes.astype(np.float32))
    with open(os.path.join(env_dir, "metadata.yaml"), "w") as fh:
        yaml.safe_dump({"box_sizes": [[0.26, 0.26]] * 6}, fh)
    # the reference reads <dataset>/<subdir>/contexts/contexts/context_XXX.pt (inference_dynamic.py:150-154)
    compat.ContextManager.save_context(torch.tensor([-0.8, -0.8]), torch.tensor([0.8, 0.8]),
                                       os.path.join(root, "data", cfg.dataset_subdir, "contexts"), cfg.dataset_subdir, 0)
    sp = make_unet_spec(cfg.state_dim, cfg.n_support_points)
    dm = DynamicGaussianDiffusionModel(model=TemporalUnetInference(n_support_points=cfg.n_support_points, state_dim=cfg.state_dim),
                                       variance_schedule=cfg.variance_schedule, n_diffusion_steps=cfg.n_diffusion_steps, predict_epsilon=True)
    full = dm.state_dict()
    for k, v in synth.make_unet_state_dict(sp, seed=0).items():
        full["model." + k] = torch.from_numpy(np.asarray(v))
    ck = os.path.join(root, "models", cfg.model_id, "checkpoints")
    os.makedirs(ck, exist_ok=True)
    torch.save(full, os.path.join(ck, "ema_model_current_state_dict.pth"))


class DynamicInference:
    def __init__(self, config: DynamicConfig):
        self.config = config
        self.device = config.device
        self.tensor_args = {'device': config.device, 'dtype': torch.float32}
        self.metrics_calculator = Metrics()
        self.context_manager = compat.ContextManager()
        self.model = None

    def run_single_experiment(self, context_idx: int, env_index: int = 0, dataset=None, obstacle_pts=None):
        """``dataset`` / ``obstacle_pts``: tests inject the reference fixture's environment and cloud; by default they come
        from the experiment directory like the reference's TrajectoryDataset item."""
        cfg = self.config
        torch.cuda.set_device(0)
        env_dir = os.path.join(cfg.dataset_path, cfg.dataset_subdir, str(env_index))
        data = compat.load_environment_dir(env_dir)
        box_centers, box_size = data['box_centers'], data['box_sizes']
        if dataset is None:      # static boxes + the pursuer sphere with its dynamics (get_dataset(..., dynamics_fn=, pursuer_pos=))
            dataset = compat.make_pursuit_env(box_centers.numpy(), box_size.numpy(), list(cfg.pursuer_pos), cfg.pursuer_radius, cfg.velocity_max)
        if obstacle_pts is None:
            obstacle_pts = data['obstacle_points']
            obstacle_pts = torch.cat([obstacle_pts[:4], obstacle_pts[torch.randint(0, 4, (2,))]], dim=0)   # 4 obstacles + 2 repeats (:142)
        n_support_points = cfg.n_support_points
        traj_normalized = torch.zeros(n_support_points, cfg.state_dim)
        start_state_pos, goal_state_pos = self.context_manager.load_context(
            os.path.join(cfg.dataset_path, cfg.dataset_subdir, 'contexts', 'contexts'), context_idx, self.device)
        diffusion_configs = dict(variance_schedule=cfg.variance_schedule, n_diffusion_steps=cfg.n_diffusion_steps, predict_epsilon=cfg.predict_epsilon)
        unet_configs = dict(state_dim=cfg.state_dim, n_support_points=n_support_points, unet_input_dim=cfg.unet_input_dim,
                            dim_mults=UNET_DIM_MULTS[cfg.unet_dim_mults_option])
        self.model = get_model(model_class=cfg.diffusion_model_class, model=TemporalUnetInference(max_rows=2 * cfg.n_samples, **unet_configs),
                               tensor_args=self.tensor_args, **diffusion_configs, **unet_configs)
        compat.load_checkpoint(self.model, cfg.trained_models_dir, cfg.model_id, use_ema=cfg.use_ema, device="cpu")
        self.model.eval()
        for p in self.model.parameters():
            p.requires_grad_(False)
        hard_conds = compat.StateGenerator.get_hard_cond_custom(torch.vstack((start_state_pos, goal_state_pos)), horizon=n_support_points,
                                                                include_velocity=cfg.include_velocity)
        context = {'dataset': dataset}
        start_state_pos = hard_conds[0][:2]; goal_state_pos = hard_conds[n_support_points - 1][:2]
        t_start_guide = ceil(cfg.start_guide_steps_fraction * self.model.n_diffusion_steps)
        sample_fn_kwargs = dict(guide=None, n_guide_steps=cfg.n_guide_steps, t_start_guide=t_start_guide, noise_std_extra_schedule_fn=lambda x: 0.5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        trajs_normalized_iters, chain_obs, chain_start = self.model.run_inference(
            context, hard_conds, n_samples=cfg.n_samples, horizon=n_support_points, return_chain=True, traj_normalized=traj_normalized,
            obstacle_pts=obstacle_pts, sample_fn=ddpm_sample_fn, **sample_fn_kwargs,
            n_diffusion_steps_without_noise=cfg.n_diffusion_steps_without_noise, max_iteration=cfg.max_replans)
        torch.cuda.synchronize(); elapsed = time.perf_counter() - t0
        chain_obs.pop()
        chain_obs = [t.cpu().detach().numpy() for t in chain_obs]
        chain_start = [np.around(t[:, :2].cpu().detach().numpy(), decimals=4) for t in chain_start]
        trajs = torch.tensor(np.stack([s.squeeze() for s in chain_start])).unsqueeze(0)           # the executed path (1, n, 2)
        ci = self.metrics_calculator.compute_collision_intensity(trajs.to(self.device), box_centers[:4].to(self.device), box_size[:4].to(self.device))
        self.last_chain = trajs_normalized_iters
        return {'chain_start': chain_start, 'chain_obs': chain_obs, 'start_state_pos': start_state_pos, 'goal_state_pos': goal_state_pos,
                'computation_time': elapsed, 'collision_intensity': bool(ci.any().item()), 'n_replans': len(chain_obs)}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--make-synthetic", metavar="DIR"); ap.add_argument("--dataset-path"); ap.add_argument("--trained-models-dir")
    ap.add_argument("--model-id", default=DynamicConfig.model_id); ap.add_argument("--context", type=int, default=0)
    ap.add_argument("--n-samples", type=int, default=DynamicConfig.n_samples); ap.add_argument("--max-replans", type=int, default=DynamicConfig.max_replans)
    args = ap.parse_args(argv)
    cfg = DynamicConfig(); cfg.n_samples, cfg.model_id, cfg.max_replans = args.n_samples, args.model_id, args.max_replans
    if args.make_synthetic:
        make_synthetic_experiment(args.make_synthetic, cfg)
        cfg.dataset_path = os.path.join(args.make_synthetic, "data"); cfg.trained_models_dir = os.path.join(args.make_synthetic, "models")
    else:
        if not (args.dataset_path and args.trained_models_dir):
            ap.error("--dataset-path and --trained-models-dir (or --make-synthetic DIR) are required")
        cfg.dataset_path, cfg.trained_models_dir = args.dataset_path, args.trained_models_dir
    runner = DynamicInference(cfg)
    metrics = runner.run_single_experiment(args.context)
    print(json.dumps({"n_replans": metrics["n_replans"], "computation_time": metrics["computation_time"],
                      "collision_intensity": metrics["collision_intensity"], "executed_states": len(metrics["chain_start"])}))
    return metrics, runner


if __name__ == "__main__":
    main()
