"""End-to-end drop-in proof: the counterpart of scripts/inference/inference_static.py run as a
program on an experiment tree in the reference's on-disk layout -- get_model -> load checkpoint -> warmup ->
run_inference -> metrics -- with the noise the reference fixture was generated with."""
import os

import numpy as np
import pytest
import torch

from oracle import ramp_oracle as O
from ramp_amd import synth
from util import GOLDEN, NoiseInjector

pytestmark = pytest.mark.gpu


def test_inference_static_entry_end_to_end(tmp_path):
    import examples.inference_static as ex
    g = np.load(f"{GOLDEN}/chain_ddpm_extra2.npz")                     # T = 25 DDPM + 2 noise-free steps, B = 4, 6 x 64 cloud
    warm = synth.make_noise((1, 4, 48, 4), seed=77)                    # warmup() draws one randn first (static.py:409)
    with NoiseInjector([warm[0]] + list(g["noise"])) as inj:
        metrics, runner = ex.main(["--make-synthetic", str(tmp_path), "--n-samples", "4", "--sampler", "ddpm",
                                   "--n-diffusion-steps", "25", "--n-steps-without-noise", "2"])
        assert inj.used == 1 + g["noise"].shape[0]
    # the tree is the reference's layout
    env_dir = tmp_path / "data" / "EnvSimple2D-RobotPointMass" / "0"
    for f in ("obstacle_points.pt", "box_centers.npy", "metadata.yaml", "contexts/context_000.pt"):
        assert (env_dir / f).exists(), f
    assert (tmp_path / "models" / "synthetic" / "checkpoints" / "ema_model_current_state_dict.pth").exists()
    final = runner.last_trajectories.cpu().numpy()
    assert metrics["n_chain_states"] == g["chain"].shape[0]
    assert np.abs(final - g["chain"][-1]).max() < 1e-4                 # the reference's own trajectories for this experiment
    # metrics of those trajectories against the oracle's restatement of scripts/inference/core/metrics.py
    boxes = synth.make_boxes(6, 2, seed=42).astype(np.float32); sizes = np.full((6, 2), 0.26, np.float32)
    ci = O.collision_intensity(g["chain"][-1], boxes, sizes)
    assert abs(metrics["collision_intensity"] - 100.0 * float(ci.mean())) < 1e-3
    free = g["chain"][-1][ci <= 0.01]
    assert metrics["n_free_trajectories"] == len(free) and metrics["success"] == int(len(free) > 0)
    if len(free):
        assert abs(metrics["path_length"] - float(O.path_length(free).mean())) < 1e-3
    assert metrics["total_time"] > 0


def test_inference_static_entry_default_mode_runs(tmp_path):
    """The script's default mode (DDIM-5 of T = 100, APF on, 5 ignored extra steps) on 64 samples."""
    import examples.inference_static as ex
    torch.manual_seed(3)
    metrics, runner = ex.main(["--make-synthetic", str(tmp_path), "--n-samples", "64", "--use-apf"])
    x = runner.last_trajectories
    assert x.shape == (64, 48, 4) and bool(torch.isfinite(x).all()) and metrics["n_chain_states"] == 6
    assert torch.equal(x[:, 0, :2], torch.tensor([-0.8, -0.8], device="cuda").expand(64, -1))
    assert torch.equal(x[:, 47, :2], torch.tensor([0.8, 0.8], device="cuda").expand(64, -1))
