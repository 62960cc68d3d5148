"""BASELINE.json's full-size workloads on the GPU, checked for correctness (not only shape / isfinite).

The delayed fp16 scaling takes its scales from maxima over ALL rows of a batch, the chunking / graph capture depend on
the row count, and the workspace is tens of GB: none of that is exercised by the B <= 512 parity tests.  At full size the
oracle cannot run, so each job embeds golden trajectories the reference itself produced (fixtures ``chain_c2.npz`` /
``chain_c3.npz`` / ``chain3d_h64_t50.npz``: the same clouds, B = 4 / 2 / 2) in its first rows -- every trajectory is
independent of its batch neighbours, so those rows must reproduce the fixture -- and checks size-independent
properties on the rest: determinism across replays, exact hard conditioning, bounded states, range guard silent.
"""
import ctypes as C
import gc

import numpy as np
import pytest
import torch

from ramp_amd import _lib, synth
import util
from util import GOLDEN, NoiseInjector, build_unet, dev

pytestmark = pytest.mark.gpu


def _free(*objs):
    for o in objs:
        m = getattr(o, "model", o)
        m._destroy_ctx()
    gc.collect()
    torch.cuda.empty_cache()


def _range_flag(u):
    flag = C.c_int32(-1)
    _lib.check(_lib.load().ramp_range_status(u.ctx(), C.byref(flag), _lib.current_stream()))
    return flag.value


def _run(dm, noise, cloud, B, return_chain=True):
    S, H = dm.state_dim, dm.model.n_support_points
    hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(S, H).items()}
    with NoiseInjector(noise, device="cuda") as inj:
        out = dm.run_inference(None, hc, n_samples=B, horizon=H, return_chain=return_chain, obstacle_pts=cloud,
                               noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=0)
        assert inj.used == len(noise)
    return out


def _noise_with_seeds(shape, gold, seed):
    """(n, B, H, S) standard normal noise on the device whose first rows are the fixture's."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    n = torch.randn(shape, device="cuda", generator=gen)
    n[:, :gold.shape[1]] = torch.from_numpy(gold).cuda()
    return [n[i] for i in range(shape[0])]


def test_config2_full_size_against_embedded_reference_trajectories():
    """BASELINE configs[1] exactly as bench.py runs it: Maze2D, B = 4096, H = 48, T = 25 DDPM, 16 x 64 = 1024-point cloud,
    APF hook for forward_t > 20, hipGraph, default fp16x3 arithmetic, one chunk of 8192 network rows."""
    from ramp_amd.models import StaticGaussianDiffusionModel
    g = np.load(f"{GOLDEN}/chain_c2.npz")
    B = 4096
    u = build_unet(4, 48, False, max_rows=2 * B)
    dm = StaticGaussianDiffusionModel(model=u, n_diffusion_steps=25, predict_epsilon=True, use_apf=True, sampler="ddpm",
                                      use_graph=True).eval().to("cuda")
    cloud = dev(g["cloud"])
    noise = _noise_with_seeds((26, B, 48, 4), g["noise"], 5)
    a = _run(dm, noise, cloud, B)
    assert _range_flag(u) == 0                                   # fp16x3 ran to the end: no fallback happened
    b = _run(dm, noise, cloud, B)                                # continues from the first job's calibration
    assert _range_flag(u) == 0
    assert torch.equal(b, _run(dm, noise, cloud, B))             # graph replay is deterministic
    print(f"config 2 full size: calibrating vs continuing job, max {float((a - b).abs().max()):.2e}")
    del b
    a = a.cpu().numpy()
    assert a.shape == (26, B, 48, 4) and np.isfinite(a).all()
    err = np.abs(a[:22, :4] - g["chain"][:22]).reshape(22, -1).max(1)
    print(f"config 2 full size: embedded golden rows, states 0..21: max {err.max():.2e}")
    assert err.max() < 1e-4                                      # free-running up to the first APF application
    # from there on the APF decisions are stiff: every remaining step teacher-forced from the reference's state
    hcb = {k: torch.from_numpy(v).cuda().unsqueeze(0).expand(B, -1) for k, v in synth.default_hard_conds(4, 48).items()}
    for j in range(21, 25):
        x_in = torch.from_numpy(a[j]).cuda(); x_in[:4] = torch.from_numpy(g["chain"][j]).cuda()
        x, _ = dm._launch(B, torch.stack([x_in, noise[j + 1]]), hcb, cloud, False, [24 - j], [1], [0.5],
                          dict(dm.apf_ddpm, passes=1), False)
        e = float(np.abs(x[:4].cpu().numpy() - g["chain"][j + 1]).max())
        assert e < 1e-4, (j, e)
    assert np.abs(a[-1]).max() <= 1.0 + 0.2                      # clamp(x0) + APF push
    hc = synth.default_hard_conds(4, 48)
    assert np.array_equal(a[:, :, 0], np.broadcast_to(hc[0], a[:, :, 0].shape))
    assert np.array_equal(a[:, :, 47], np.broadcast_to(hc[47], a[:, :, 47].shape))
    # rows are independent: the golden rows alone (B = 4, another context) land within rounding of their full-size selves
    u4 = build_unet(4, 48, False, max_rows=8)
    d4 = StaticGaussianDiffusionModel(model=u4, n_diffusion_steps=25, predict_epsilon=True, use_apf=True, sampler="ddpm",
                                      use_graph=True).eval().to("cuda")
    small = _run(d4, [n[:4].contiguous() for n in noise], cloud, 4).cpu().numpy()
    assert np.abs(small[:22] - a[:22, :4]).max() < 1e-4
    _free(dm, d4)


def _golden_rows_exact_fp32(cls, S, H, T, g, n_rows):
    """The embedded golden rows alone (B = n_rows) on the exact-fp32 MFMA mode -- the one mode whose arithmetic is the reference's -- free-running:
    the bars of the full-size fp16x3 job are asserted on it too; rows are independent, so B does not matter."""
    u = build_unet(S, H, True, max_rows=2 * n_rows, gemm_mode="fp32")
    dm = cls(model=u, n_diffusion_steps=T, predict_epsilon=True, use_graph=True).eval().to("cuda")
    c = _run(dm, [torch.from_numpy(n).cuda() for n in g["noise"]], dev(g["cloud"]), n_rows).cpu().numpy()
    _free(dm)
    return c


def test_config3_full_size_against_embedded_reference_trajectories():
    """BASELINE configs[2]: Maze3D, B = 4096, H = 48, S = 6, T = 25 DDPM, w = 5.75, 20 x 200 = 4000-point cloud."""
    from ramp_amd.models import GaussianDiffusionModel3d
    g = np.load(f"{GOLDEN}/chain_c3.npz")
    B = 4096
    u = build_unet(6, 48, True, max_rows=2 * B)
    dm = GaussianDiffusionModel3d(model=u, n_diffusion_steps=25, predict_epsilon=True, use_graph=True).eval().to("cuda")
    cloud = dev(g["cloud"])
    noise = _noise_with_seeds((26, B, 48, 6), g["noise"], 6)
    a = _run(dm, noise, cloud, B)
    assert _range_flag(u) == 0
    b = _run(dm, noise, cloud, B)                                # continues from the first job's calibration
    assert _range_flag(u) == 0
    assert torch.equal(b, _run(dm, noise, cloud, B))
    del b
    a = a.cpu().numpy()
    assert np.isfinite(a).all() and np.abs(a[-1]).max() <= 1.0
    # w = 5.75 amplifies rounding ~12x per step: the free-running chain is comparable at 5e-4 (the float64 truth is 1e-4
    # from the reference's own fp32 chain, tests/test_gpu_sampler.py::test_chain3d_...), every step teacher-forced at 1e-4
    err = np.abs(a[:, :2] - g["chain"]).reshape(26, -1).max(1)
    truth = util.oracle64_chain("chain_c3", 6, 48, 25, 5.75)
    e_ref = np.abs(g["chain"] - truth).max(); e_gpu = np.abs(a[:, :2] - truth).max()
    print(f"config 3 full size: embedded golden rows free-running max {err.max():.2e}; vs float64 truth: reference {e_ref:.2e}, HIP {e_gpu:.2e} (ratio {e_gpu / e_ref:.2f})")
    # free-running w = 5.75 chains are chaotic: two builds of this library land 1.2e-4 (ratio 0.5) from the float64 truth and 1.25e-4 / 2.2e-4
    # from the reference (whose own chain is 2.4e-4 from the truth).  Bars: as close to the truth as 3 x the reference's own distance and
    # 2 x the worst measured; the sharp accuracy statement is per step (tests/test_gpu_sampler.py, assert_as_accurate_as_the_reference)
    assert e_gpu < 3 * e_ref and e_gpu < 2.5e-4 and err.max() < 4.5e-4
    c32 = _golden_rows_exact_fp32(GaussianDiffusionModel3d, 6, 48, 25, g, 2)
    e32 = np.abs(c32 - truth).max(); r32 = np.abs(c32 - g["chain"]).max()
    print(f"   the same rows on the exact-fp32 MFMA mode: vs reference {r32:.2e}, vs float64 truth {e32:.2e} (ratio {e32 / e_ref:.2f})")
    assert e32 < 3 * e_ref and e32 < 2.5e-4 and r32 < 4.5e-4                # the SAME bars: they describe the chain, not the emulation
    hcb = {k: torch.from_numpy(v).cuda().unsqueeze(0).expand(B, -1) for k, v in synth.default_hard_conds(6, 48).items()}
    tf = []                                                      # EVERY step from the reference's own previous state
    for j in range(25):
        x_in = torch.from_numpy(a[j]).cuda(); x_in[:2] = torch.from_numpy(g["chain"][j]).cuda()
        x, _ = dm._launch(B, torch.stack([x_in, noise[j + 1]]), hcb, cloud, False, [24 - j], [0], [0.5], None, False)
        tf.append(float(np.abs(x[:2].cpu().numpy() - g["chain"][j + 1]).max()))
    print("config 3 full size: teacher-forced per step " + " ".join(f"{e:.1e}" for e in tf))
    assert max(tf) < 1e-4, tf
    hc = synth.default_hard_conds(6, 48)
    assert np.array_equal(a[-1][:, 0], np.broadcast_to(hc[0], (B, 6))) and np.array_equal(a[-1][:, 47], np.broadcast_to(hc[47], (B, 6)))
    _free(dm)


def test_config5_per_gpu_shard_full_size():
    """One GPU's shard of BASELINE configs[4]: Maze3D, B = 8192 (of 65536 over 8 GPUs), H = 64, T = 50, 40 x 200 = 8000-point
    cloud, hipGraph of 50 steps x 16384 network rows (the ~110 GB workspace).  The golden rows come from the H = 64 /
    T = 50 reference chain (its own small cloud): the scene enters only through the latent, so the job runs on THAT
    latent; a second job on the 8000-point cloud checks the encoder + properties at the configured size."""
    from ramp_amd.models import GaussianDiffusionModel3d
    g = np.load(f"{GOLDEN}/chain3d_h64_t50.npz")
    B = 8192
    u = build_unet(6, 64, True, max_rows=2 * B)
    dm = GaussianDiffusionModel3d(model=u, n_diffusion_steps=50, predict_epsilon=True, use_graph=True).eval().to("cuda")
    noise = _noise_with_seeds((51, B, 64, 6), g["noise"], 7)
    a = _run(dm, noise, dev(g["cloud"]), B).cpu().numpy()
    assert _range_flag(u) == 0
    assert a.shape == (51, B, 64, 6) and np.isfinite(a).all() and np.abs(a[-1]).max() <= 1.0
    err = np.abs(a[:, :2] - g["chain"]).reshape(51, -1).max(1)
    truth = util.oracle64_chain("chain3d_h64_t50", 6, 64, 50, 5.75)
    e_ref = np.abs(g["chain"] - truth).max(); e_gpu = np.abs(a[:, :2] - truth).max()
    print(f"config 5 shard: embedded golden rows free-running max {err.max():.2e} (final {err[-1]:.2e}); vs float64 truth: reference {e_ref:.2e}, HIP {e_gpu:.2e} (ratio {e_gpu / e_ref:.2f})")
    # (chaotic, see config 3: measured 5.6e-4 from the truth (the reference itself: 4.3e-4), 6.6e-4 from the reference; T = 50 steps of 12x amplification)
    assert e_gpu < 3 * e_ref and e_gpu < 1.1e-3 and err.max() < 1.3e-3
    c32 = _golden_rows_exact_fp32(GaussianDiffusionModel3d, 6, 64, 50, g, 2)
    e32 = np.abs(c32 - truth).max(); r32 = np.abs(c32 - g["chain"]).max()
    print(f"   the same rows on the exact-fp32 MFMA mode: vs reference {r32:.2e}, vs float64 truth {e32:.2e} (ratio {e32 / e_ref:.2f})")
    assert e32 < 3 * e_ref and e32 < 1.1e-3 and r32 < 1.3e-3                # the SAME bars: they describe the chain, not the emulation
    hcb = {k: torch.from_numpy(v).cuda().unsqueeze(0).expand(B, -1) for k, v in synth.default_hard_conds(6, 64).items()}
    tf = []                                                      # EVERY step from the reference's own previous state
    for j in range(50):
        x_in = torch.from_numpy(a[j]).cuda(); x_in[:2] = torch.from_numpy(g["chain"][j]).cuda()
        x, _ = dm._launch(B, torch.stack([x_in, noise[j + 1]]), hcb, dev(g["cloud"]), False, [49 - j], [0], [0.5], None, False)
        tf.append(float(np.abs(x[:2].cpu().numpy() - g["chain"][j + 1]).max()))
    print("config 5 shard: teacher-forced per step " + " ".join(f"{e:.1e}" for e in tf))
    assert max(tf) < 1e-4, tf
    del a
    big = dev(synth.make_cloud(40, 200, 3, seed=42))
    x1 = _run(dm, noise, big, B, return_chain=False)
    assert _range_flag(u) == 0
    x2 = _run(dm, noise, big, B, return_chain=False)
    assert _range_flag(u) == 0
    assert torch.equal(x2, _run(dm, noise, big, B, return_chain=False))
    assert bool(torch.isfinite(x1).all()) and float(x1.abs().max()) <= 1.0
    hc = synth.default_hard_conds(6, 64)
    assert torch.equal(x1[:, 0], torch.from_numpy(hc[0]).cuda().expand(B, -1))
    assert torch.equal(x1[:, 63], torch.from_numpy(hc[63]).cuda().expand(B, -1))
    _free(dm)


def test_config4_full_size_replanning_graph_equals_eager_launch():
    """BASELINE configs[3]: the dynamic planner at B = 8192 candidates, 10 high-level DDIM steps + 3 replans x 5 steps, a
    1024-point pursuer cloud re-sampled every replan, one captured graph per replan (fp16x3 with the calibration carried
    from replan to replan).  At this size the check is structural: the captured-graph run and the same launches issued
    eagerly must agree bit for bit (same kernels, same order, same delayed-scaling tables), twice in a row; the executed
    states are pinned in every later plan; start / goal conditions hold; the pursuer moved."""
    from ramp_amd import compat
    from ramp_amd.models import DynamicGaussianDiffusionModel
    B, H, S = 8192, 48, 4
    cloud = dev(synth.make_cloud(16, 64, 2, seed=42))
    boxes = synth.make_boxes(16, 2, seed=42)
    hcn = synth.default_hard_conds(S, H)
    hc = {k: torch.from_numpy(v).cuda().unsqueeze(0).expand(B, -1).contiguous() for k, v in hcn.items()}
    u = build_unet(S, H, False, max_rows=2 * B)
    outs = []
    for use_graph in (True, False, True):
        dm = DynamicGaussianDiffusionModel(model=u, n_diffusion_steps=100, predict_epsilon=True, use_graph=use_graph).eval().to("cuda")
        dm.apf_dynamic = dict(dm.apf_dynamic, points_per_obstacle=1024)
        ctx = {"dataset": compat.make_pursuit_env(boxes, np.full((len(boxes), 2), 0.26), [0.6, 0.55])}
        torch.manual_seed(11); np.random.seed(11)
        x, chain, obs, start = dm.ddim_p_sample_loop((B, H, S), hc, context=ctx, return_chain=True, obstacle_pts=cloud,
                                                     max_iteration=3)
        assert _range_flag(u) == 0
        outs.append((x.cpu().numpy(), chain.cpu().numpy(), np.stack([o.cpu().numpy() for o in obs])))
    x, chain, obs = outs[0]
    for other in outs[1:]:
        assert np.array_equal(x, other[0]) and np.array_equal(chain, other[1]) and np.array_equal(obs, other[2])
    assert np.isfinite(chain).all() and chain.shape[0] == 1 and chain.shape[2:] == (H, S)
    n_plans = chain.shape[1]                       # high-level plan + one per replan
    assert 2 <= n_plans <= 4
    assert np.array_equal(x[-1], chain[0, 0, -1])                       # every replan keeps the plan's goal waypoint
    assert np.array_equal(x[0, :2], hcn[0][:2]) and np.all(x[0, 2:] == 0)
    for k in range(2, n_plans):                                         # plan k keeps the states executed before it
        assert np.array_equal(chain[0, k, :k - 1], chain[0, k - 1, :k - 1])
    assert np.abs(obs[-1] - obs[0]).max() > 0 or n_plans == 2           # the pursuer moved between replans
    _free(u)
