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


def test_inference3d_entry_end_to_end(tmp_path):
    """The counterpart of scripts/inference/inference3d.py (:19-156) run as a program on an experiment tree in the
    reference's layout, Maze3D: get_model(GaussianDiffusionModel3d) -> checkpoint -> hard conditions through the limits
    normaliser -> run_inference(return_chain=True), with the config-3 cloud (20 obstacles x 200 points) and the noise of the
    reference fixture chain_c3.npz (two independent n_samples = 1 reference runs = one n_samples = 2 run here).
    Free-running w = 5.75 chain: within 5e-4 of the reference's (the documented bar for these chains, test_gpu_sampler.py)."""
    import examples.inference3d as ex
    g = np.load(f"{GOLDEN}/chain_c3.npz")
    ex.main.synthetic_cloud = g["cloud"]
    try:
        with NoiseInjector(list(g["noise"])) as inj:
            out, chain = ex.main(["--make-synthetic", str(tmp_path), "--n-samples", "2", "--n-diffusion-steps", "25"])
            assert inj.used == g["noise"].shape[0]
    finally:
        del ex.main.synthetic_cloud
    env_dir = tmp_path / "data" / "EnvSmall3D" / "0"
    for f in ("obstacle_points.pt", "box_centers.npy", "sphere_centers.npy", "metadata.yaml"):
        assert (env_dir / f).exists(), f
    chain = chain.cpu().numpy()
    assert chain.shape == g["chain"].shape and out["n_chain_states"] == 26
    err = np.abs(chain - g["chain"]).reshape(26, -1).max(1)
    print(f"inference3d e2e: final {err[-1]:.2e} max {err.max():.2e}")
    assert err.max() < 5e-4
    assert out["start_error"] < 1e-6 and out["goal_error"] < 1e-6            # hard conditioning survives the un-normalisation


def test_inference_dynamic_entry_end_to_end(tmp_path):
    """The counterpart of scripts/inference/inference_dynamic.py (:105-275) run as a program: experiment tree in the
    reference's layout -> get_model(DynamicGaussianDiffusionModel) -> checkpoint -> context file -> run_inference (one
    ramp_sample job + one captured ramp_replan graph per replan) -> executed path + collision intensity.  With the
    environment, cloud, noise and numpy seed of the reference fixture replan_chain.npz the planner reproduces every batch
    the reference handed to its cost selection, every selected index and collision mask."""
    import examples.inference_dynamic as ex
    from util import StopReplan, make_fake_pursuit_env
    g = np.load(f"{GOLDEN}/replan_chain.npz")
    K = int(g["n_iter"]); B = g["noise"].shape[1]
    cfg = ex.DynamicConfig(); cfg.n_samples = B
    ex.make_synthetic_experiment(str(tmp_path), cfg)
    # the fixture's start / goal (hard0 / hardN) as the experiment's context file
    ex.compat.ContextManager.save_context(torch.from_numpy(g["hard0"][:2]), torch.from_numpy(g["hardN"][:2]),
                                          str(tmp_path / "data" / cfg.dataset_subdir / "contexts"), cfg.dataset_subdir, 0)
    cfg.dataset_path = str(tmp_path / "data"); cfg.trained_models_dir = str(tmp_path / "models")
    runner = ex.DynamicInference(cfg)
    log_env = []
    dataset, _sphere = make_fake_pursuit_env(stop_at=K, log=log_env)
    np.random.seed(23)
    import ramp_amd.diffusion as D
    orig_init = D.DynamicGaussianDiffusionModel.__init__

    def logging_init(self, *a, **k):
        orig_init(self, *a, **k)
        self.replan_log = []

    D.DynamicGaussianDiffusionModel.__init__ = logging_init
    try:
        with NoiseInjector(list(g["noise"])):
            with pytest.raises(StopReplan):                                    # the fixture run was cut after K replans
                runner.run_single_experiment(0, dataset=dataset, obstacle_pts=torch.from_numpy(g["cloud"]))
    finally:
        D.DynamicGaussianDiffusionModel.__init__ = orig_init
    log = runner.model.replan_log
    assert len(log) == int(g["n_cost"]) and len(log_env) == int(g["n_env"])
    worst = 0.0
    for j, e in enumerate(log):
        worst = max(worst, float(np.abs(e["batch"].cpu().numpy() - g[f"cost{j}/trajs"]).max()))
        assert e["idx"] == int(g[f"cost{j}/idx"]) and np.array_equal(e["free"].cpu().numpy(), g[f"cost{j}/free"]), j
    print(f"inference_dynamic e2e: worst ranked batch {worst:.2e}")
    assert worst < 2e-4


def test_inference_dynamic_entry_runs_to_the_end(tmp_path):
    """The script's own environment (static boxes, pursuer dynamics of scripts/inference/core/utils.py:85-137) for a few
    replans: executed states are pinned, the path starts at the context's start, outputs have the reference's structure."""
    import examples.inference_dynamic as ex
    torch.manual_seed(5); np.random.seed(5)
    metrics, runner = ex.main(["--make-synthetic", str(tmp_path), "--n-samples", "64", "--max-replans", "4"])
    assert metrics["n_replans"] == len(metrics["chain_obs"]) <= 4 and len(metrics["chain_start"]) == metrics["n_replans"] + 2
    assert np.allclose(metrics["chain_start"][0], [[-0.8, -0.8]])
    assert runner.last_chain.shape[1:] == (1, 48, 4) and bool(torch.isfinite(runner.last_chain).all())
    assert (tmp_path / "data" / "EnvSimple2D-RobotPointMass" / "contexts" / "contexts" / "context_000.pt").exists()
