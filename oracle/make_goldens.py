#!/usr/bin/env python3
"""Capture golden input/output vectors from the reference itself (runs in the BUILD container only).

Imports ``mpd.models`` from /root/reference (read-only mount, never present on the GPU box),
loads the repo's own seeded synthetic weights (ramp_amd/synth.py) into it, injects noise by
patching ``torch.randn`` / ``torch.randn_like`` and writes small ``.npz`` fixtures to
tests/golden/.  The fixtures hold DATA only (inputs, expected outputs, a few intermediate
activations / gradients), never reference source text.

    python oracle/make_goldens.py            # regenerates every fixture (≈2-4 min on 8 cores)

What is captured:
  schedule_T{25,50,100}.npz   the 12 schedule buffers
  unet2d_h48.npz              forward_no_energy, eps, per-module outputs + output-grads (N=4 rows)
  unet3d_h48.npz / h64        same for the 3-D net (N=2 rows: cond, uncond)
  scene_latents.npz           2-D 6x64 / 16x64 clouds, 3-D 5x50 / 20x200 clouds
  chain_ddpm_*.npz            full T=25 DDPM chains, B=4, with / without APF, with extra no-noise steps
  chain_ddim_*.npz            DDIM-5 of T=100 chains, with / without APF
  chain3d_ddpm.npz            3-D DDPM T=25: B independent n_samples=1 runs stacked
  unet2d_h40.npz / chain_ddpm_h40.npz   a horizon other than 48 / 64 (levels 40, 20, 10, 5): taps and a DDPM chain
  chain3d_h64_t50.npz         3-D DDPM, H=64, T=50 (the shape of BASELINE config 5): B independent n_samples=1 runs stacked
  chain_c2.npz / chain_c3.npz the clouds of BASELINE configs 2 and 3 (16x64 2-D with the APF hook; 20x200 3-D), B = 4 / 2 golden
                              trajectories that the full-size GPU tests embed in their B = 4096 batches
  compose_static.npz          compose=True: one p_mean_variance_compose, a DDPM T=25 chain (use_apf=True: no hook on this
                              path), the DDIM-8 of T=100 + APF chain on the 10-obstacle union cloud
  compose_3d.npz              3-D compose (w1 = w2 = 5) DDPM T=25: B independent n_samples=1 runs stacked
  apf_cases.npz               avoidance() in/out pairs (hits, no-hit early-out, window clipped at ends)
  cost_cases.npz              compute_collision_with_pointcloud / compute_trajectory_costs
  boundary_cases.npz          public helper methods of the sampler class + predict_epsilon=False (one p_mean_variance, a DDPM chain)
  unet2d_h48_outlier.npz      the 2-D score evaluation with outlier-channel weights (8 rows of every to_out / ff.net.2 x 2^9)
  chain_ddpm_outlier.npz      the T = 25 DDPM chain on those weights (fp32 reference run + its float64 twin)
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RAMP_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from ramp_amd import synth  # noqa: E402
from ramp_amd.spec import make_unet_spec, unet_param_shapes, SCHEDULE_BUFFERS  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from mpd.models import (TemporalUnetInference, UNET_DIM_MULTS, StaticGaussianDiffusionModel,  # noqa: E402
                            GaussianDiffusionModel3d, DynamicGaussianDiffusionModel)
    from mpd.models.diffusion_models import APFhelper_dynamic as ref_apf_dyn  # noqa: E402
    from mpd.models.diffusion_models.sample_functions import ddpm_sample_fn  # noqa: E402
    from mpd.models.diffusion_models.APFhelper import ObstacleField, avoidance  # noqa: E402
    from mpd.models.diffusion_models import cost as ref_cost  # noqa: E402
    from mpd.models.diffusion_models.obstacle_encoder import ObstacleEncoderSet  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build_unet(state_dim, horizon, obstacle_3d, seed=0, outliers=False):
    sp = make_unet_spec(state_dim, horizon, obstacle_3d=obstacle_3d)
    sd = synth.make_unet_state_dict(sp, seed=seed)
    if outliers:
        sd = synth.add_outlier_channels(sd)
    m = quiet(TemporalUnetInference, n_support_points=horizon, state_dim=state_dim, unet_input_dim=32,
              dim_mults=UNET_DIM_MULTS[1], obstacle_3d=obstacle_3d)
    ref_sd = m.state_dict()
    assert set(ref_sd.keys()) == set(sd.keys()), "spec.py key name mismatch with reference"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m.eval()
    for p in m.parameters():
        p.requires_grad_(False)
    return m, sp, sd


class NoiseInjector:
    """Replace torch.randn / randn_like by a pre-generated list (consumed in call order)."""

    def __init__(self, tensors):
        self.q = list(tensors)
        self.used = 0

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like

        def randn(*shape, **kw):
            if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
                shape = tuple(shape[0])
            t = self.q[self.used]
            self.used += 1
            assert tuple(t.shape) == tuple(shape), (t.shape, shape)
            return t.clone()

        def randn_like(x, **kw):
            t = self.q[self.used]
            self.used += 1
            assert t.shape == x.shape
            return t.clone()

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------------------------------------
def gen_schedules(unet2d):
    for T in (25, 50, 100):
        dm = quiet(StaticGaussianDiffusionModel, model=unet2d, variance_schedule="exponential",
                   n_diffusion_steps=T, predict_epsilon=True)
        save(f"schedule_T{T}.npz", **{k: getattr(dm, k).numpy() for k in SCHEDULE_BUFFERS})


def hook_modules(m, names):
    """forward hooks capturing module outputs and (through tensor hooks) grads w.r.t. those outputs."""
    outs, grads, handles = OrderedDict(), OrderedDict(), []
    mods = dict(m.named_modules())
    for n in names:
        def fh(mod, inp, out, n=n):
            outs[n] = out.detach().clone()
            if out.requires_grad:
                out.register_hook(lambda g, n=n: grads.__setitem__(n, g.detach().clone()))
        handles.append(mods[n].register_forward_hook(fh))
    return outs, grads, handles


def module_names(sp):
    names = []
    for k, lv in enumerate(sp.downs):
        names += [f"downs.{k}.0", f"downs.{k}.1", f"downs.{k}.3"]
        if lv.resample:
            names.append(f"downs.{k}.4")
    names += ["mid_block1", "mid_attention", "mid_block2"]
    for k, lv in enumerate(sp.ups):
        names += [f"ups.{k}.0", f"ups.{k}.1", f"ups.{k}.3", f"ups.{k}.4"]
    return names


def cl(t):  # (N,C,L) -> channels-last (N,L,C)
    return t.permute(0, 2, 1).contiguous().numpy()


def gen_unet(tag, m, sp, cloud, n_rows, t_val, seed):
    """One score-net evaluation with per-module taps. cloud (No,Np,D)."""
    H, S = sp.horizon, sp.state_dim
    x = torch.from_numpy(synth.make_noise((n_rows, H, S), seed=seed))
    t = torch.full((n_rows,), t_val, dtype=torch.long)
    pts = torch.from_numpy(cloud)[None].repeat(n_rows, 1, 1, 1)
    m.reset_cache()
    lat_full = m.scene_encoder(pts[:1])[0].detach().numpy()
    names = module_names(sp)
    outs, grads, handles = hook_modules(m, names)
    m.reset_cache()
    eps = m(x, t, None, obstacle_pts=pts).detach()
    for h in handles:
        h.remove()
    m.reset_cache()
    with torch.no_grad():
        f = m.forward_no_energy(x, t, obstacle_pts=pts)
    # time embedding
    temb = m.time_mlp(t).detach().numpy()
    arrs = dict(x=x.numpy(), t=t.numpy(), cloud=cloud, latent=lat_full, f=f.numpy(), eps=eps.numpy(), temb=temb)
    for n in names:
        arrs["out/" + n] = cl(outs[n])
        if n in grads:
            arrs["gout/" + n] = cl(grads[n])
    save(f"unet{tag}.npz", **arrs)


def gen_scene_latents(m2, m3):
    arrs = {}
    c = synth.make_cloud(6, 64, 2, seed=42)
    arrs["cloud2d_6x64"] = c
    arrs["lat2d_6x64"] = m2.scene_encoder(torch.from_numpy(c)[None])[0].detach().numpy()
    enc16 = ObstacleEncoderSet(num_obstacles=16)
    enc16.load_state_dict(m2.scene_encoder.state_dict())
    enc16.eval()
    c = synth.make_cloud(16, 64, 2, seed=43)
    arrs["cloud2d_16x64"] = c
    arrs["lat2d_16x64"] = enc16(torch.from_numpy(c)[None])[0].detach().numpy()
    for (no, npnt, sd) in ((5, 50, 44), (20, 200, 45)):
        c = synth.make_cloud(no, npnt, 3, seed=sd)
        arrs[f"cloud3d_{no}x{npnt}"] = c
        arrs[f"lat3d_{no}x{npnt}"] = m3.scene_encoder(torch.from_numpy(c)[None])[0].detach().numpy()
    save("scene_latents.npz", **arrs)


def run_static(unet, sp, T, B, cloud, noise, ddim, use_apf, n_without_noise=0):
    dm = quiet(StaticGaussianDiffusionModel, model=unet, variance_schedule="exponential", n_diffusion_steps=T,
               predict_epsilon=True, compose=False, use_apf=use_apf)
    dm.eval()
    dm.ddim = ddim
    unet.reset_cache()
    hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(sp.state_dim, sp.horizon).items()}
    pts = torch.from_numpy(cloud)
    with NoiseInjector([torch.from_numpy(n) for n in noise]) as inj:
        chain = dm.run_inference(None, hc, n_samples=B, horizon=sp.horizon, return_chain=True,
                                 traj_normalized=torch.zeros(sp.horizon, sp.state_dim), obstacle_pts=pts,
                                 sample_fn=ddpm_sample_fn, guide=None, n_guide_steps=1, t_start_guide=7,
                                 noise_std_extra_schedule_fn=lambda x: 0.5,
                                 n_diffusion_steps_without_noise=n_without_noise)
        used = inj.used
    return chain.detach().numpy(), used


def gen_chains(m2, sp2):
    B, H, S = 4, sp2.horizon, sp2.state_dim
    cloud = synth.make_cloud(6, 64, 2, seed=42)
    m2.reset_cache()
    latent = m2.scene_encoder(torch.from_numpy(cloud)[None])[0].detach().numpy()
    # DDPM T=25
    for tag, apf, nwn in (("plain", False, 0), ("apf", True, 0), ("extra2", False, 2)):
        n_steps = 25 + nwn
        noise = synth.make_noise((n_steps + 1, B, H, S), seed=1234)
        chain, used = run_static(m2, sp2, 25, B, cloud, noise, ddim=False, use_apf=apf, n_without_noise=nwn)
        assert used == n_steps + 1 and chain.shape == (n_steps + 1, B, H, S), (used, chain.shape)
        save(f"chain_ddpm_{tag}.npz", chain=chain, noise=noise, cloud=cloud, latent=latent, T=25,
             n_without_noise=nwn, use_apf=apf)
    # DDIM-5 of T=100 (script default mode)
    for tag, apf in (("plain", False), ("apf", True)):
        noise = synth.make_noise((1, B, H, S), seed=4321)
        chain, used = run_static(m2, sp2, 100, B, cloud, noise, ddim=True, use_apf=apf, n_without_noise=5)
        assert used == 1 and chain.shape == (6, B, H, S)
        save(f"chain_ddim_{tag}.npz", chain=chain, noise=noise, cloud=cloud, latent=latent, T=100, K=5, use_apf=apf)


def gen_chain3d(m3, sp3):
    H, S, T, B = sp3.horizon, sp3.state_dim, 25, 2
    cloud = synth.make_cloud(5, 50, 3, seed=44)
    m3.reset_cache()
    latent = m3.scene_encoder(torch.from_numpy(cloud)[None])[0].detach().numpy()
    noise = synth.make_noise((T + 1, B, H, S), seed=777)
    chains = []
    for b in range(B):
        dm = quiet(GaussianDiffusionModel3d, model=m3, variance_schedule="exponential", n_diffusion_steps=T,
                   predict_epsilon=True, compose=False, use_apf=False)
        dm.eval()
        m3.reset_cache()
        hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(S, H).items()}
        with NoiseInjector([torch.from_numpy(noise[j, b:b + 1]) for j in range(T + 1)]) as inj:
            chain = dm.run_inference(None, hc, n_samples=1, horizon=H, return_chain=True,
                                     traj_normalized=torch.zeros(H, S), obstacle_pts=torch.from_numpy(cloud),
                                     sample_fn=ddpm_sample_fn, guide=None, n_guide_steps=1, t_start_guide=7,
                                     noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=0)
            assert inj.used == T + 1
        chains.append(chain.detach().numpy())
    chain = np.concatenate(chains, axis=1)
    save("chain3d_ddpm.npz", chain=chain, noise=noise, cloud=cloud, latent=latent, T=T, w=5.75)


def gen_chain3d_h64(m3b, sp3b):
    """BASELINE config 5's shape: 3-D, H = 64, T = 50 DDPM (diffusion_model_3d.py:185-218), B independent runs."""
    H, S, T, B = sp3b.horizon, sp3b.state_dim, 50, 2
    cloud = synth.make_cloud(8, 40, 3, seed=45)
    m3b.reset_cache()
    latent = m3b.scene_encoder(torch.from_numpy(cloud)[None])[0].detach().numpy()
    noise = synth.make_noise((T + 1, B, H, S), seed=778)
    chains = []
    for b in range(B):
        dm = quiet(GaussianDiffusionModel3d, model=m3b, variance_schedule="exponential", n_diffusion_steps=T,
                   predict_epsilon=True, compose=False, use_apf=False)
        dm.eval()
        m3b.reset_cache()
        hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(S, H).items()}
        with NoiseInjector([torch.from_numpy(noise[j, b:b + 1]) for j in range(T + 1)]) as inj:
            chain = dm.run_inference(None, hc, n_samples=1, horizon=H, return_chain=True,
                                     traj_normalized=torch.zeros(H, S), obstacle_pts=torch.from_numpy(cloud),
                                     sample_fn=ddpm_sample_fn, guide=None, n_guide_steps=1, t_start_guide=7,
                                     noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=0)
            assert inj.used == T + 1
        chains.append(chain.detach().numpy())
    chain = np.concatenate(chains, axis=1)
    assert chain.shape == (T + 1, B, H, S)
    save("chain3d_h64_t50.npz", chain=chain, noise=noise, cloud=cloud, latent=latent, T=T, w=5.75)


def gen_horizon40(m40, sp40):
    """A horizon other than the drivers' 48 / 64: n_support_points = 40 (levels of 40, 20, 10 and 5 tokens) -- the
    reference takes any multiple of 8 (UnetInference.py:42-56).  One tapped score evaluation and a free-running DDPM chain."""
    cloud = synth.make_cloud(6, 64, 2, seed=42)
    gen_unet("2d_h40", m40, sp40, cloud, 4, 11, seed=13)
    T, B = 25, 3
    noise = synth.make_noise((T + 1, B, 40, 4), seed=14)
    chain, used = run_static(m40, sp40, T, B, cloud, noise, ddim=False, use_apf=False)
    assert used == T + 1
    save("chain_ddpm_h40.npz", chain=chain, noise=noise, cloud=cloud, T=T)


def gen_fullsize_seeds(m2, sp2, m3, sp3):
    """Golden trajectories for the full-size property tests: BASELINE config 2's cloud (16 x 64 points: the 2-D encoder
    built with num_obstacles=16, same weights -- SURVEY Q4) with the DDPM APF hook, and config 3's 20 x 200 3-D cloud."""
    B, H, S = 4, sp2.horizon, sp2.state_dim
    cloud = synth.make_cloud(16, 64, 2, seed=42)
    enc6 = m2.scene_encoder
    enc16 = ObstacleEncoderSet(num_obstacles=16)
    enc16.load_state_dict(enc6.state_dict())
    enc16.eval()
    for p in enc16.parameters():
        p.requires_grad_(False)
    m2.scene_encoder = enc16
    try:
        m2.reset_cache()
        latent = enc16(torch.from_numpy(cloud)[None])[0].detach().numpy()
        noise = synth.make_noise((26, B, H, S), seed=1234)
        chain, used = run_static(m2, sp2, 25, B, cloud, noise, ddim=False, use_apf=True)
        assert used == 26
        save("chain_c2.npz", chain=chain, noise=noise, cloud=cloud, latent=latent, T=25, use_apf=True)
    finally:
        m2.scene_encoder = enc6
        m2.reset_cache()
    H, S, T, B = sp3.horizon, sp3.state_dim, 25, 2
    cloud = synth.make_cloud(20, 200, 3, seed=42)
    m3.reset_cache()
    latent = m3.scene_encoder(torch.from_numpy(cloud)[None])[0].detach().numpy()
    noise = synth.make_noise((T + 1, B, H, S), seed=777)
    chains = []
    for b in range(B):
        dm = quiet(GaussianDiffusionModel3d, model=m3, variance_schedule="exponential", n_diffusion_steps=T,
                   predict_epsilon=True, compose=False, use_apf=False)
        dm.eval()
        m3.reset_cache()
        hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(S, H).items()}
        with NoiseInjector([torch.from_numpy(noise[j, b:b + 1]) for j in range(T + 1)]) as inj:
            chain = dm.run_inference(None, hc, n_samples=1, horizon=H, return_chain=True,
                                     traj_normalized=torch.zeros(H, S), obstacle_pts=torch.from_numpy(cloud),
                                     sample_fn=ddpm_sample_fn, guide=None, n_guide_steps=1, t_start_guide=7,
                                     noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=0)
            assert inj.used == T + 1
        chains.append(chain.detach().numpy())
    m3.reset_cache()
    save("chain_c3.npz", chain=np.concatenate(chains, axis=1), noise=noise, cloud=cloud, latent=latent, T=T, w=5.75)


def gen_compose(m2, sp2):
    """compose=True on the static wrapper (diffusion_model_static.py:188-229, 259-333): scene A + scene B + uncond."""
    H, S, B = sp2.horizon, sp2.state_dim, 3
    clouds = np.stack([synth.make_cloud(6, 64, 2, seed=1), synth.make_cloud(6, 64, 2, seed=2)])
    pts = torch.from_numpy(clouds)
    arrs = dict(clouds=clouds)
    # (1) one p_mean_variance_compose, DDIM return signature (model_mean, ..., x_recon, e_comb)
    dm = quiet(StaticGaussianDiffusionModel, model=m2, variance_schedule="exponential", n_diffusion_steps=25,
               predict_epsilon=True, compose=True, use_apf=False)
    dm.eval()
    m2.reset_cache()
    x = synth.make_noise((B, H, S), seed=3)
    t = torch.full((B,), 9, dtype=torch.long)
    dm.ddim = True
    mean, _, _, x0, ec = dm.p_mean_variance_compose(torch.from_numpy(x.copy()), None, None, t, traj_normalized=None,
                                                    obstacle_pts=pts, compose=True)
    arrs.update(pmv_x=x, pmv_t=9, pmv_mean=mean.detach().numpy(), pmv_x0=x0.detach().numpy(), pmv_ecomb=ec.detach().numpy())
    hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(S, H).items()}
    kw = dict(horizon=H, return_chain=True, traj_normalized=torch.zeros(H, S), obstacle_pts=pts, sample_fn=ddpm_sample_fn,
              guide=None, n_guide_steps=1, t_start_guide=7, noise_std_extra_schedule_fn=lambda x: 0.5,
              n_diffusion_steps_without_noise=0)
    # (2) DDPM T=25 with use_apf=True: ddpm_sample_fn -> p_mean_variance_compose, which has no APF hook
    dm = quiet(StaticGaussianDiffusionModel, model=m2, variance_schedule="exponential", n_diffusion_steps=25,
               predict_epsilon=True, compose=True, use_apf=True)
    dm.eval(); dm.ddim = False
    m2.reset_cache()
    noise = synth.make_noise((26, B, H, S), seed=2345)
    with NoiseInjector([torch.from_numpy(n) for n in noise]) as inj:
        chain = dm.run_inference(None, hc, n_samples=B, **kw).detach().numpy()
        assert inj.used == 26 and chain.shape == (26, B, H, S)
    arrs.update(ddpm_noise=noise, ddpm_chain=chain)
    # (3) DDIM-8 of T=100 with the APF hook (forward_t >= 2, three passes, union cloud of 6 + 4 obstacles)
    dm = quiet(StaticGaussianDiffusionModel, model=m2, variance_schedule="exponential", n_diffusion_steps=100,
               predict_epsilon=True, compose=True, use_apf=True)
    dm.eval()
    assert dm.ddim and dm.ddim_num_inference_steps == 8
    m2.reset_cache()
    noise = synth.make_noise((1, B, H, S), seed=5432)
    with NoiseInjector([torch.from_numpy(n) for n in noise]) as inj:
        chain = dm.run_inference(None, hc, n_samples=B, **kw).detach().numpy()
        assert inj.used == 1 and chain.shape == (9, B, H, S)
    arrs.update(ddim_noise=noise, ddim_chain=chain)
    m2.reset_cache()
    save("compose_static.npz", **arrs)


def gen_compose3d(m3, sp3):
    """3-D compose (diffusion_model_3d.py:163-182, w1 = w2 = 5; rows [scene A, scene B, uncond], valid for one sample):
    B independent n_samples=1 DDPM T=25 runs stacked."""
    H, S, T, B = sp3.horizon, sp3.state_dim, 25, 2
    clouds = np.stack([synth.make_cloud(5, 50, 3, seed=44), synth.make_cloud(5, 50, 3, seed=46)])
    noise = synth.make_noise((T + 1, B, H, S), seed=779)
    chains = []
    for b in range(B):
        dm = quiet(GaussianDiffusionModel3d, model=m3, variance_schedule="exponential", n_diffusion_steps=T,
                   predict_epsilon=True, compose=True, use_apf=False)
        dm.eval()
        m3.reset_cache()
        hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(S, H).items()}
        with NoiseInjector([torch.from_numpy(noise[j, b:b + 1]) for j in range(T + 1)]) as inj:
            chain = dm.run_inference(None, hc, n_samples=1, horizon=H, return_chain=True,
                                     traj_normalized=torch.zeros(H, S), obstacle_pts=torch.from_numpy(clouds),
                                     sample_fn=ddpm_sample_fn, guide=None, n_guide_steps=1, t_start_guide=7,
                                     noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=0)
            assert inj.used == T + 1
        chains.append(chain.detach().numpy())
    m3.reset_cache()
    chain = np.concatenate(chains, axis=1)
    lats = m3.scene_encoder(torch.from_numpy(clouds)).detach().numpy()
    m3.reset_cache()
    save("compose_3d.npz", chain=chain, noise=noise, clouds=clouds, latents=lats, T=T, w1=5.0, w2=5.0)


def gen_boundary(m2, sp2):
    """The sampler classes' public helpers and the constructor default predict_epsilon=False
    (diffusion_model_static.py:96-147, 149-186, 259-333): in/out pairs of predict_start_from_noise,
    predict_noise_from_start, q_posterior, deep_repeat_tensor, one static ddim_p_sample step (with / without the APF hook),
    one p_mean_variance and a T = 25 DDPM chain of a predict_epsilon=False wrapper (the network output IS x0)."""
    H, S, B = sp2.horizon, sp2.state_dim, 3
    cloud = synth.make_cloud(6, 64, 2, seed=42)
    pts = torch.from_numpy(cloud)
    arrs = dict(cloud=cloud)
    x = synth.make_noise((B, H, S), seed=31); z = synth.make_noise((B, H, S), seed=32)
    t = torch.full((B,), 9, dtype=torch.long)
    arrs.update(x=x, z=z, t=9)
    for pe in (True, False):
        dm = quiet(StaticGaussianDiffusionModel, model=m2, variance_schedule="exponential", n_diffusion_steps=25,
                   predict_epsilon=pe)
        dm.eval()
        tag = "eps" if pe else "x0"
        arrs[f"psn_{tag}"] = dm.predict_start_from_noise(torch.from_numpy(x), t, torch.from_numpy(z)).numpy()
        arrs[f"pns_{tag}"] = dm.predict_noise_from_start(torch.from_numpy(x), t, torch.from_numpy(z)).numpy()
        qm, qv, qlv = dm.q_posterior(x_start=torch.from_numpy(z), x_t=torch.from_numpy(x), t=t)
        arrs[f"q_mean_{tag}"] = qm.numpy(); arrs[f"q_var_{tag}"] = qv.numpy(); arrs[f"q_logvar_{tag}"] = qlv.numpy()
        m2.reset_cache()
        dm.ddim = True
        mean, _, _, x0, ec = dm.p_mean_variance(torch.from_numpy(x.copy()), None, None, t,
                                                traj_normalized=torch.zeros(H, S), obstacle_pts=pts.unsqueeze(0))
        arrs[f"pmv_mean_{tag}"] = mean.detach().numpy(); arrs[f"pmv_x0_{tag}"] = x0.detach().numpy()
        arrs[f"pmv_ecomb_{tag}"] = ec.detach().numpy()
    dm = quiet(StaticGaussianDiffusionModel, model=m2, variance_schedule="exponential", n_diffusion_steps=25, predict_epsilon=True)
    xr, tr, trj, obr = dm.deep_repeat_tensor(torch.from_numpy(x), torch.arange(B), torch.from_numpy(z), pts.unsqueeze(0), 2)
    arrs.update(rep_x=xr.numpy(), rep_t=tr.numpy(), rep_traj=trj.numpy(), rep_obst_shape=np.asarray(obr.shape))
    # one static DDIM step of T = 100 / K = 5 at t = 40 (forward_t = 2: the APF hook's first step), with and without the hook
    hc = {k: torch.from_numpy(v)[None].repeat(B, 1) for k, v in synth.default_hard_conds(S, H).items()}
    # (the state: what the reference's own DDIM loop holds before that step -- chain_ddim_apf.npz, same cloud, state 2.  On a random state the
    # hook moves x0 by O(1) through its collision tests and amplifies a 1e-7 perturbation of x0 to 1e-4 .. 1e-3: not a parity fixture)
    xs = np.load(os.path.join(OUT, "chain_ddim_apf.npz"))["chain"][2][:B].copy()
    arrs["ddim_x"] = xs
    for apf in (False, True):
        dm = quiet(StaticGaussianDiffusionModel, model=m2, variance_schedule="exponential", n_diffusion_steps=100,
                   predict_epsilon=True, compose=False, use_apf=apf)
        dm.eval()
        m2.reset_cache()
        out = dm.ddim_p_sample(torch.from_numpy(xs.copy()), hc, None, torch.full((B,), 40, dtype=torch.long), pts.unsqueeze(0),
                               traj_normalized=torch.zeros(H, S), forward_t=2, eta=0.0, use_clipped_model_output=True)
        arrs["ddim_out_apf" if apf else "ddim_out"] = out.detach().numpy()
    # predict_epsilon=False (the constructor default): a DDPM chain (T = 25: the exponential schedule of a T = 5 wrapper has
    # beta_4 = 1 + 1 ulp and NaN posterior coefficients in the reference itself)
    T, Bc = 25, 2
    dm = quiet(StaticGaussianDiffusionModel, model=m2, variance_schedule="exponential", n_diffusion_steps=T,
               predict_epsilon=False, compose=False, use_apf=False)
    dm.eval(); dm.ddim = False
    m2.reset_cache()
    noise = synth.make_noise((T + 1, Bc, H, S), seed=34)
    hc1 = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(S, H).items()}
    with NoiseInjector([torch.from_numpy(n) for n in noise]) as inj:
        chain = dm.run_inference(None, hc1, n_samples=Bc, horizon=H, return_chain=True, traj_normalized=torch.zeros(H, S),
                                 obstacle_pts=pts, sample_fn=ddpm_sample_fn, guide=None, n_guide_steps=1, t_start_guide=7,
                                 noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=0)
        assert inj.used == T + 1
    arrs.update(x0_chain=chain.detach().numpy(), x0_noise=noise)
    m2.reset_cache()
    save("boundary_cases.npz", **arrs)


def gen_outlier_unet():
    """One tapped score evaluation of the 2-D net whose transformer output projections carry outlier channels
    (synth.add_outlier_channels: 8 rows of every attn1.to_out / ff.net.2 scaled by 2^9) -- trained-transformer statistics."""
    m, sp, _ = build_unet(4, 48, False, outliers=True)
    gen_unet("2d_h48_outlier", m, sp, synth.make_cloud(6, 64, 2, seed=42), 4, 7, seed=10)


def gen_outlier_chain():
    """The T = 25 DDPM chain (B = 4, the cloud / hard conditions of chain_ddpm_plain) on the outlier-channel weights: where the delayed
    (previous-evaluation) operand scales of the fp16x3 mode meet operands that move along a chain.  ``chain`` is the
    reference's fp32 run, ``chain64`` the same run with model, noise and cloud in float64: the yardstick for what ANY fp32 evaluation can
    reproduce free-running."""
    m, sp, _ = build_unet(4, 48, False, outliers=True)
    B, H, S = 4, sp.horizon, sp.state_dim
    cloud = synth.make_cloud(6, 64, 2, seed=42)
    noise = synth.make_noise((26, B, H, S), seed=1234)
    chain, used = run_static(m, sp, 25, B, cloud, noise, ddim=False, use_apf=False)
    assert used == 26 and chain.shape == (26, B, H, S)
    torch.set_default_dtype(torch.float64)
    try:
        m.double()
        dm = quiet(StaticGaussianDiffusionModel, model=m, variance_schedule="exponential", n_diffusion_steps=25,
                   predict_epsilon=True, compose=False, use_apf=False).double()
        dm.eval(); dm.ddim = False
        m.reset_cache()
        hc = {k: torch.from_numpy(v).double() for k, v in synth.default_hard_conds(S, H).items()}
        with NoiseInjector([torch.from_numpy(n).double() for n in noise]):
            chain64 = dm.run_inference(None, hc, n_samples=B, horizon=H, return_chain=True, traj_normalized=torch.zeros(H, S),
                                       obstacle_pts=torch.from_numpy(cloud).double(), sample_fn=ddpm_sample_fn,
                                       noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=0).detach().numpy()
    finally:
        torch.set_default_dtype(torch.float32)
    d = np.abs(chain - chain64).reshape(26, -1).max(1)
    print(f"    outlier chain: reference fp32 vs its float64 twin, final {d[-1]:.2e} max {d.max():.2e}; max |x| {np.abs(chain).max():.2f}")
    save("chain_ddpm_outlier.npz", chain=chain, chain64=chain64, noise=noise, cloud=cloud, T=25)


def gen_apf():
    arrs = {}
    cloud = synth.make_cloud(6, 64, 2, seed=42).reshape(-1, 2)
    g = np.random.Generator(np.random.PCG64(99))
    cases = {}
    # (a) random trajectories over the workspace: many hits
    cases["rand"] = (g.uniform(-1, 1, size=(3, 48, 4)).astype(np.float32), cloud, 0.07, 0.1, 5)
    # (b) straight line start->goal, window 7
    lin = np.linspace(-0.8, 0.8, 48, dtype=np.float32)
    tr = np.zeros((2, 48, 4), np.float32)
    tr[0, :, 0] = lin; tr[0, :, 1] = lin
    tr[1, :, 0] = lin; tr[1, :, 1] = -lin
    cases["line"] = (tr, cloud, 0.07, 0.1, 7)
    # (c) no hit: cloud far away -> early-out returns the input unchanged
    cases["nohit"] = (g.uniform(-0.2, 0.2, size=(2, 48, 4)).astype(np.float32), cloud + 5.0, 0.07, 0.1, 5)
    # (d) hits only at the first / last waypoints: window clipped at both ends
    tr = np.full((2, 48, 4), 3.0, np.float32)
    tr[0, 0, :2] = cloud[10] + np.float32(0.01)
    tr[0, 1, :2] = cloud[11] - np.float32(0.02)
    tr[1, 47, :2] = cloud[200] + np.float32(0.03)
    cases["ends"] = (tr, cloud, 0.07, 0.1, 7)
    # (e) larger cloud / threshold
    cloud16 = synth.make_cloud(16, 64, 2, seed=43).reshape(-1, 2)
    cases["big"] = (g.uniform(-1, 1, size=(4, 48, 4)).astype(np.float32), cloud16, 0.1, 0.2, 7)
    for name, (traj, cl_, thr, strength, win) in cases.items():
        field = ObstacleField(cl_, distance_threshold=thr)
        out = avoidance(torch.from_numpy(traj.copy()), field, avoidance_window=win, avoidance_strength=strength)
        arrs[f"{name}/traj"] = traj
        arrs[f"{name}/cloud"] = cl_
        arrs[f"{name}/params"] = np.array([thr, strength, win], np.float64)
        arrs[f"{name}/out"] = out.numpy()
        print(f"    apf case {name}: changed elements = {(out.numpy() != traj).sum()}")
    save("apf_cases.npz", **arrs)


def gen_cost():
    cloud = synth.make_cloud(6, 64, 2, seed=42)
    g = np.random.Generator(np.random.PCG64(5))
    base = np.linspace(-0.8, 0.8, 48, dtype=np.float32)
    trajs = np.zeros((10, 48, 4), np.float32)
    for b in range(10):
        if b % 2 == 0:   # diagonal-ish paths through the obstacle field (mostly colliding)
            trajs[b, :, 0] = base + 0.05 * g.standard_normal(48).astype(np.float32)
            trajs[b, :, 1] = base * (1 if b % 4 == 0 else -1) + 0.3 * np.sin(np.linspace(0, 3, 48)).astype(np.float32) * b / 8
        else:            # paths hugging the workspace border (x = -0.97), free of every box
            trajs[b, :, 0] = -0.97 + 0.005 * g.standard_normal(48).astype(np.float32) * b
            trajs[b, :, 1] = base * (0.5 + 0.05 * b)
        trajs[b, 1:, 2:] = np.diff(trajs[b, :, :2], axis=0)
    arrs = dict(trajs=trajs, cloud=cloud)
    for thr in (0.02, 0.05, 0.1):
        mask = ref_cost.compute_collision_with_pointcloud(torch.from_numpy(trajs), torch.from_numpy(cloud), thr)
        arrs[f"mask_{thr}"] = mask.numpy()
        print(f"    collision mask thr={thr}: {mask.numpy().astype(int)}")
    best, best_cost, total, free, idx = quiet(ref_cost.compute_trajectory_costs, torch.from_numpy(trajs),
                                              torch.from_numpy(cloud), collision_threshold=0.05)
    assert best is not None
    arrs["best"] = best.numpy(); arrs["total_costs"] = total.numpy(); arrs["best_index"] = int(idx)
    arrs["free_mask"] = free.numpy()
    arrs["path_length"] = ref_cost.compute_path_length(torch.from_numpy(trajs)).numpy()
    arrs["smoothness"] = ref_cost.compute_smoothness(torch.from_numpy(trajs)).numpy()
    save("cost_cases.npz", **arrs)


def gen_dynamic(m2, sp2):
    """Dynamic (pursuit-evasion) wrapper: CFG with the blocked row layout (w = 2.5; reference quirk Q1: the net
    masks odd GLOBAL rows) for even and odd batch sizes, and the per-trajectory dynamic APF (static + pursuer)."""
    arrs = {}
    cloud = synth.make_cloud(6, 64, 2, seed=42)
    dm = quiet(DynamicGaussianDiffusionModel, model=m2, variance_schedule="exponential", n_diffusion_steps=100,
               predict_epsilon=True)
    dm.eval()
    for B in (4, 3):
        x = torch.from_numpy(synth.make_noise((B, 48, 4), seed=30 + B))
        t = torch.full((B,), 40, dtype=torch.long)
        pts = torch.from_numpy(cloud)[None].repeat(B, 1, 1, 1)
        m2.reset_cache()
        mean, _, _, x0, ec = dm.p_mean_variance(x, None, None, t, traj_normalized=torch.zeros(B, 48, 4), obstacle_pts=pts)
        arrs[f"pmv{B}/x"] = x.numpy(); arrs[f"pmv{B}/ecomb"] = ec.detach().numpy()
        arrs[f"pmv{B}/x0"] = x0.detach().numpy(); arrs[f"pmv{B}/mean"] = mean.detach().numpy()
    arrs["cloud"] = cloud
    # dynamic APF: seeded numpy RNG so the generated point sets can be captured
    np.random.seed(7)
    centers = np.array([[-0.3, 0.2], [0.35, -0.25], [0.1, 0.55], [-0.5, -0.5]])
    sizes = np.full((4, 2), 0.26)
    pursuer = np.array([0.05, 0.0])
    field = ref_apf_dyn.ObstacleField(centers, sizes, lambda t, sp, rg=False, bi=None: (pursuer, 0.1), 64,
                                      distance_threshold=0.2, distance_threshold_pred=0.5)
    field.update_dynamic(0, None)
    arrs["apf/static_points"] = field.static_obstacle_points
    arrs["apf/dynamic_points"] = np.asarray(field.dynamic_kdtree.data)
    g = np.random.Generator(np.random.PCG64(17))
    lin = np.linspace(-0.8, 0.8, 48, dtype=np.float32)
    trajs = np.zeros((4, 48, 4), np.float32)
    for b in range(4):
        trajs[b, :, 0] = lin + 0.03 * g.standard_normal(48).astype(np.float32)
        trajs[b, :, 1] = (0.9 - 0.45 * b) * lin + 0.03 * g.standard_normal(48).astype(np.float32)
    trajs[3] += 5.0                                            # far from everything: nothing may change
    goal = torch.tensor([0.8, 0.8, 0.0, 0.0])
    arrs["apf/traj"] = trajs; arrs["apf/goal"] = goal.numpy()
    out_s = np.stack([ref_apf_dyn.avoidance(torch.from_numpy(trajs[b].copy()), field, is_dynamic=False,
                                            avoidance_window=8, avoidance_strength=0.15,
                                            avoidance_strength_pred=0.15).numpy() for b in range(4)])
    out_d = np.stack([ref_apf_dyn.avoidance(torch.from_numpy(trajs[b].copy()), field, is_dynamic=True,
                                            avoidance_window=5, avoidance_strength=0.15, avoidance_strength_pred=0.15,
                                            affected_states=48, goal_state=goal).numpy() for b in range(4)])
    arrs["apf/out_static"] = out_s; arrs["apf/out_dynamic"] = out_d
    arrs["apf/params"] = np.array([0.2, 0.5, 0.15, 0.15, 8, 5], np.float64)   # thr_static, thr_pred, strengths, windows
    print(f"    dynamic apf: static changed {(out_s != trajs).sum()}, dynamic changed {(out_d != trajs).sum()}")
    # sm() velocity smoothing
    s1 = torch.from_numpy(synth.make_noise((5, 4), seed=41)); s2 = torch.from_numpy(synth.make_noise((5, 4), seed=42))
    arrs["sm/s1"] = s1.numpy(); arrs["sm/s2"] = s2.numpy(); arrs["sm/out"] = dm.sm(s1, s2).numpy()
    save("dynamic_cases.npz", **arrs)


FAKE_BOX_CENTRES = [[-0.3, 0.2], [0.35, -0.25], [0.1, 0.55], [-0.5, -0.5], [0.7, -0.7], [-0.7, 0.7]]


class _StopReplan(Exception):
    pass


def make_fake_pursuit_env(stop_at=None, log=None):
    """Stand-in for context['dataset'].env as the dynamic planner touches it: obj_fixed_list[0].fields[0] has
    .centers / .sizes (boxes), obj_extra_list[0].fields[0] has .centers (1,2) / .radii (1,) and
    update_centers(t, current_state): a deterministic pursuer stepping 0.05 toward the mean evader position.
    tests/util.py holds the identical definition for the HIP side."""
    from types import SimpleNamespace as NS

    class Sphere:
        def __init__(self):
            self.centers = torch.tensor([[0.6, 0.55]], dtype=torch.float32)
            self.radii = torch.tensor([0.1], dtype=torch.float32)

        def update_centers(self, t, current_state):
            if log is not None:
                log.append((int(t), current_state.detach().cpu().numpy().copy()))
            if stop_at is not None and t >= stop_at:
                raise _StopReplan()
            tgt = current_state.detach().cpu().float().mean(dim=0)[:2]
            d = tgt - self.centers[0]
            n = float(torch.linalg.norm(d))
            step = d * (0.05 / n) if n > 0.05 else d
            self.centers = (self.centers[0] + step).unsqueeze(0)

    boxes = NS(centers=torch.tensor(FAKE_BOX_CENTRES), sizes=torch.full((6, 2), 0.16))
    sphere = Sphere()
    env = NS(obj_fixed_list=[NS(fields=[boxes])], obj_extra_list=[NS(fields=[sphere])])
    return NS(env=env), sphere


def gen_replan(m2, sp2):
    """Receding-horizon planner (diffusion_model_dynamic.py:495-624) with a fake env, seeded numpy RNG, recorded
    torch noise; the reference's cost selection is instrumented from outside to log each batch it ranks."""
    import mpd.models.diffusion_models.diffusion_model_dynamic as ref_dyn
    B, H, S, K = 6, 48, 4, 4
    boxes_c = np.array(FAKE_BOX_CENTRES)
    np.random.seed(11)
    cloud = np.stack([ref_apf_dyn.generate_box_points(c, (0.16, 0.16), 64) for c in boxes_c]).astype(np.float32)
    dm = quiet(DynamicGaussianDiffusionModel, model=m2, variance_schedule="exponential", n_diffusion_steps=100,
               predict_epsilon=True)
    dm.eval()
    noises = [torch.from_numpy(synth.make_noise((B, H, S), seed=500 + i)) for i in range(K + 2)]
    log_env, log_cost = [], []
    dataset, sphere = make_fake_pursuit_env(stop_at=K, log=log_env)
    orig = ref_dyn.compute_trajectory_costs

    def logged(trajs, pts, **kw):
        out = orig(trajs, pts, **kw)
        log_cost.append((trajs.detach().numpy().copy(), np.asarray(pts).reshape(-1, 2).copy(),
                         -1 if out[4] is None else int(out[4]), out[3].numpy().copy(),
                         None if out[0] is None else out[0].detach().numpy().copy()))
        return out

    ref_dyn.compute_trajectory_costs = logged
    hard = {0: torch.tensor([-0.8, -0.8, 0.0, 0.0]).repeat(B, 1), H - 1: torch.tensor([0.8, 0.8, 0.0, 0.0]).repeat(B, 1)}
    m2.reset_cache()
    np.random.seed(23)
    try:
        with NoiseInjector(noises) as inj:
            quiet(dm.ddim_p_sample_loop, (B, H, S), hard, context={'dataset': dataset}, return_chain=True,
                  traj_normalized=torch.zeros(B, H, S), obstacle_pts=torch.from_numpy(cloud))
            used = inj.used
    except _StopReplan:
        used = inj.used
    finally:
        ref_dyn.compute_trajectory_costs = orig
    arrs = {"cloud": cloud, "noise": np.stack([n.numpy() for n in noises[:used]]), "n_iter": np.array(K),
            "hard0": hard[0][0].numpy(), "hardN": hard[H - 1][0].numpy()}
    for j, (tr, pts, idx, free, best) in enumerate(log_cost):
        arrs[f"cost{j}/trajs"] = tr; arrs[f"cost{j}/npts"] = np.array(pts.shape[0]); arrs[f"cost{j}/idx"] = np.array(idx)
        arrs[f"cost{j}/free"] = free
        if best is not None:
            arrs[f"cost{j}/best"] = best
    for j, (t, st) in enumerate(log_env):
        arrs[f"env{j}/t"] = np.array(t); arrs[f"env{j}/state"] = st
    arrs["n_cost"] = np.array(len(log_cost)); arrs["n_env"] = np.array(len(log_env))
    arrs["pursuer_final"] = sphere.centers.numpy()
    # the same run in float64 (model, noise, cloud, default dtype): how far the reference's OWN fp32 run is from exact arithmetic at every
    # logged batch -- the yardstick for a free-running chain that passes through APF pushes and re-selection
    n32 = len(log_cost)
    log_cost64, log_env64 = [], []
    dataset64, sphere64 = make_fake_pursuit_env(stop_at=K, log=log_env64)

    def logged64(trajs, pts, **kw):
        out = orig(trajs, pts, **kw)
        log_cost64.append((trajs.detach().numpy().copy(), -1 if out[4] is None else int(out[4]), out[3].numpy().copy()))
        return out

    ref_dyn.compute_trajectory_costs = logged64
    torch.set_default_dtype(torch.float64)
    m2.double(); dm.double()
    m2.reset_cache()
    np.random.seed(23)
    try:
        with NoiseInjector([n.double() for n in noises]):
            quiet(dm.ddim_p_sample_loop, (B, H, S), {k: v.double() for k, v in hard.items()}, context={'dataset': dataset64}, return_chain=True,
                  traj_normalized=torch.zeros(B, H, S), obstacle_pts=torch.from_numpy(cloud).double())
    except _StopReplan:
        pass
    finally:
        ref_dyn.compute_trajectory_costs = orig
        torch.set_default_dtype(torch.float32)
        m2.float(); dm.float(); m2.reset_cache()
    assert len(log_cost64) == n32
    for j, (tr64, idx64, free64) in enumerate(log_cost64):
        assert tr64.dtype == np.float64 and idx64 == log_cost[j][2] and np.array_equal(free64, log_cost[j][3]), j      # the same plan
        arrs[f"cost{j}/trajs64"] = tr64
    print("    replan: the reference's fp32 run vs its float64 twin, per ranked batch:",
          [f"{np.abs(log_cost[j][0] - log_cost64[j][0]).max():.2e}" for j in range(n32)])
    print(f"    replan: {len(log_cost)} selections (idx {[c[2] for c in log_cost]}, free {[int(c[3].sum()) for c in log_cost]}), "
          f"{len(log_env)} pursuer updates, {used} noise draws")
    save("replan_chain.npz", **arrs)


def gen_metrics():
    """Metrics class of scripts/inference/core/metrics.py on border-hugging trajectories (some inside boxes)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_metrics", os.path.join(REF, "scripts/inference/core/metrics.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    M = mod.Metrics()
    g = np.random.Generator(np.random.PCG64(77))
    B, H = 24, 48
    lin = np.linspace(-0.9, 0.9, H, dtype=np.float32)
    tr = np.zeros((B, H, 4), np.float32)
    for b in range(B):
        tr[b, :, 0] = lin + 0.05 * g.standard_normal(H).astype(np.float32)
        tr[b, :, 1] = (g.uniform(-1, 1)) * lin + 0.05 * g.standard_normal(H).astype(np.float32) + g.uniform(-0.3, 0.3)
        tr[b, :, 2:] = 0.3 * g.standard_normal((H, 2)).astype(np.float32)
    centers = np.array([[-0.3, 0.2], [0.35, -0.25], [0.1, 0.55], [-0.5, -0.5], [0.0, 0.0]], np.float32)
    sizes = np.array([[0.2, 0.2], [0.16, 0.3], [0.1, 0.1], [0.25, 0.12], [0.08, 0.08]], np.float32)
    t = torch.from_numpy(tr)
    ci = M.compute_collision_intensity(t, torch.from_numpy(centers), torch.from_numpy(sizes))
    res = M.trajectory_success_and_metrics(t, ci, threshold=0.01)
    arrs = {"traj": tr, "centers": centers, "sizes": sizes, "intensity": ci.numpy(),
            "path_length": M.compute_path_length(t).numpy(), "smoothness": M.compute_smoothness(t).numpy(),
            "variance_all": float(M.compute_variance_waypoints(t)),
            "success": res["success"], "collision_intensity_pct": res["collision_intensity"],
            "n_free": res["n_free_trajectories"], "free_path_length": np.float64(res["path_length"] or np.nan),
            "free_path_length_std": np.float64(res["path_length_std"] or np.nan),
            "free_variance": np.float64(res["waypoint_variance"] if res["waypoint_variance"] is not None else np.nan)}
    print(f"    metrics: intensity>0 for {(ci > 0).sum().item()} of {B}, n_free {res['n_free_trajectories']}, var {arrs['variance_all']:.6f}")
    save("metrics_cases.npz", **arrs)


def gen_compat():
    """Pursuer dynamics (scripts/inference/core/utils.py:85-137) and LimitsNormalizer (normalization.py:144-167) outputs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_utils", os.path.join(REF, "scripts/inference/core/utils.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    fn, vel = mod.DynamicsGenerator.create_pursuit_dynamics(0.5)
    g = np.random.Generator(np.random.PCG64(5))
    ts = np.arange(12); prev = g.uniform(-1, 1, (12, 1, 2)); robot = g.uniform(-1, 1, (12, 1, 2))
    prev[3] = robot[3]                                            # zero distance branch
    prev[4] = [[0.999, -0.999]]                                   # clipping
    out = np.stack([fn(int(t), prev[i], robot[i], vel) for i, t in enumerate(ts)])
    # (mpd.datasets' package __init__ imports gitpython, absent here: load the one module file)
    nspec = importlib.util.spec_from_file_location("ref_norm", os.path.join(REF, "mpd/datasets/normalization.py"))
    nmod = importlib.util.module_from_spec(nspec); nspec.loader.exec_module(nmod)
    x = torch.from_numpy(g.uniform(-2, 3, (7, 5, 4)).astype(np.float32))
    n = nmod.LimitsNormalizer(x.reshape(-1, 4))                   # mins / maxs over the flattened data (normalization.py:90-93)
    z = n.normalize(x); xb = n.unnormalize(z * 1.2)
    hc = mod.StateGenerator.get_hard_cond_custom(torch.tensor([[0.1, -0.2], [0.5, 0.6], [0.7, 0.8]]), horizon=48)
    save("compat_cases.npz", dyn_t=ts, dyn_prev=prev, dyn_robot=robot, dyn_vel=vel, dyn_out=out, norm_x=x.numpy(),
         norm_mins=np.asarray(n.mins), norm_maxs=np.asarray(n.maxs), norm_z=z.numpy(), norm_back=xb.numpy(),
         hc0=hc[0].numpy(), hc47=hc[47].numpy())


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "dynamic":
        m2, sp2, _ = build_unet(4, 48, False)
        gen_dynamic(m2, sp2); return
    if len(sys.argv) > 1 and sys.argv[1] == "compat":
        gen_compat(); return
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":
        gen_metrics(); return
    if len(sys.argv) > 1 and sys.argv[1] == "replan":
        m2, sp2, _ = build_unet(4, 48, False)
        gen_replan(m2, sp2); return
    if len(sys.argv) > 1 and sys.argv[1] == "compose":
        m2, sp2, _ = build_unet(4, 48, False)
        gen_compose(m2, sp2)
        m3, sp3, _ = build_unet(6, 48, True)
        gen_compose3d(m3, sp3); return
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":
        m2, sp2, _ = build_unet(4, 48, False)
        m3, sp3, _ = build_unet(6, 48, True)
        gen_fullsize_seeds(m2, sp2, m3, sp3); return
    if len(sys.argv) > 1 and sys.argv[1] == "config5":
        m3b, sp3b, _ = build_unet(6, 64, True)
        gen_chain3d_h64(m3b, sp3b); return
    if len(sys.argv) > 1 and sys.argv[1] == "h40":
        m40, sp40, _ = build_unet(4, 40, False)
        gen_horizon40(m40, sp40); return
    if len(sys.argv) > 1 and sys.argv[1] == "boundary":
        m2, sp2, _ = build_unet(4, 48, False)
        gen_boundary(m2, sp2); return
    if len(sys.argv) > 1 and sys.argv[1] == "outlier":
        gen_outlier_unet(); return
    if len(sys.argv) > 1 and sys.argv[1] == "outlier_chain":
        gen_outlier_chain(); return
    if len(sys.argv) > 1 and sys.argv[1] == "cost":
        gen_cost(); return
    if len(sys.argv) > 1 and sys.argv[1] == "apf":
        gen_apf(); return
    print("building reference models with synthetic weights ...")
    m2, sp2, _ = build_unet(4, 48, False)
    m3, sp3, _ = build_unet(6, 48, True)
    m3b, sp3b, _ = build_unet(6, 64, True)
    print("schedules"); gen_schedules(m2)
    print("unet taps")
    gen_unet("2d_h48", m2, sp2, synth.make_cloud(6, 64, 2, seed=42), 4, 7, seed=10)
    gen_unet("3d_h48", m3, sp3, synth.make_cloud(5, 50, 3, seed=44), 2, 3, seed=11)
    gen_unet("3d_h64", m3b, sp3b, synth.make_cloud(5, 50, 3, seed=44), 2, 20, seed=12)
    print("scene latents"); gen_scene_latents(m2, m3)
    print("apf"); gen_apf()
    print("cost"); gen_cost()
    print("chains 2-D"); gen_chains(m2, sp2)
    print("chain 3-D"); gen_chain3d(m3, sp3)
    print("chain 3-D H=64 T=50"); gen_chain3d_h64(m3b, sp3b)
    print("compose"); gen_compose(m2, sp2); gen_compose3d(m3, sp3)
    print("full-size seeds"); gen_fullsize_seeds(m2, sp2, m3, sp3)
    print("dynamic"); gen_dynamic(m2, sp2)
    print("replan"); gen_replan(m2, sp2)
    print("metrics"); gen_metrics()
    print("compat"); gen_compat()
    print("horizon 40"); m40, sp40, _ = build_unet(4, 40, False); gen_horizon40(m40, sp40)
    print("boundary helpers / predict_epsilon=False"); gen_boundary(m2, sp2)
    print("outlier-channel weights"); gen_outlier_unet()
    print("done")


if __name__ == "__main__":
    main()
