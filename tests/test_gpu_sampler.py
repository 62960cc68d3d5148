"""Sampler-loop parity on the GPU: whole reverse-diffusion chains through the reference-compatible
run_inference(), against chains captured from the reference itself with injected noise."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ramp_oracle as O
from ramp_amd import synth
import util
from util import GOLDEN, NoiseInjector, build_unet, dev, rel, weights

pytestmark = pytest.mark.gpu


def make_static(T, use_apf=False, sampler="ddpm", use_graph=True, max_rows=64, gemm_mode="default", launch_plan=None):
    from ramp_amd.models import StaticGaussianDiffusionModel
    u = build_unet(4, 48, False, max_rows=max_rows, gemm_mode=gemm_mode, launch_plan=launch_plan)
    dm = StaticGaussianDiffusionModel(model=u, variance_schedule="exponential", n_diffusion_steps=T,
                                      predict_epsilon=True, compose=False, use_apf=use_apf, sampler=sampler,
                                      use_graph=use_graph)
    return dm.eval().to("cuda")


def run(dm, g, B, n_without_noise=0):
    hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(dm.state_dim, dm.model.n_support_points).items()}
    with NoiseInjector(list(g["noise"])) as inj:
        chain = dm.run_inference(None, hc, n_samples=B, horizon=dm.model.n_support_points, return_chain=True,
                                 traj_normalized=None, obstacle_pts=dev(g["cloud"]),
                                 sample_fn=None, guide=None, n_guide_steps=1, t_start_guide=7,
                                 noise_std_extra_schedule_fn=lambda x: 0.5,
                                 n_diffusion_steps_without_noise=n_without_noise)
        used = inj.used
    return chain.cpu().numpy(), used


@pytest.mark.parametrize("tag,nwn", [("plain", 0), ("extra2", 2)])
@pytest.mark.parametrize("graph", [True, False])
def test_ddpm_chain_free_running(tag, nwn, graph):
    """T=25 DDPM from x_T, every one of the T+1 states, vs the reference chain (BASELINE.json: 1e-4 fp32;
    the reference's own fp32-vs-fp64 drift on this chain is 3.5e-5)."""
    g = np.load(f"{GOLDEN}/chain_ddpm_{tag}.npz")
    dm = make_static(25, use_graph=graph)
    chain, used = run(dm, g, 4, nwn)
    assert used == g["noise"].shape[0] and chain.shape == g["chain"].shape
    err = np.abs(chain - g["chain"]).reshape(chain.shape[0], -1).max(1)
    print(f"ddpm {tag} graph={graph}: final {err[-1]:.2e} max {err.max():.2e}")
    assert err.max() < 1e-4
    assert np.array_equal(chain[:, :, 0], np.broadcast_to(synth.default_hard_conds(4, 48)[0], chain[:, :, 0].shape))
    assert np.array_equal(chain[:, :, 47], np.broadcast_to(synth.default_hard_conds(4, 48)[47], chain[:, :, 47].shape))


@pytest.mark.parametrize("mode", ["fp32", "bf16x6", "fp16x3", "fp16x3-fusedff", "fp16x3-ffx", "fp16x3-atk", "fp16x3-tkc", "fp16x3-tkw", "fp16x3-m32"])
def test_ddpm_chain_every_gemm_mode(mode):
    """The same reference chain in each GEMM mode: exact fp32 MFMA (v_mfma_f32_32x32x2_f32), bf16x6 (three bf16 planes,
    six products) and fp16x3 (two scaled fp16 planes, three products; its first evaluation calibrates in bf16x6),
    the last one also with every feed-forward through the fused FF1 -> GEGLU -> FF2 kernel, and through the token-owning
    fused forward + backward pair of ffx.hip (by default only the large launches take them; forced through the launch plan)."""
    g = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")
    mode, plan = util.split_mode(mode)
    chain, _ = run(make_static(25, gemm_mode=mode, launch_plan=plan), g, 4)
    err = np.abs(chain - g["chain"]).max()
    print(f"ddpm plain {mode} mode: max {err:.2e}")
    assert err < 1e-4


@pytest.mark.parametrize("fixture", ["chain_ddpm_plain", "chain_ddim_apf"])
def test_shared_prefix_against_row_by_row_evaluation(fixture):
    """Inside a sampling job the CFG rows of a trajectory share the part of the network they have in common (level-0
    residual blocks, first transformer down to its first cross-attention constant) and its input gradient is taken once,
    on the weighted sum of the rows' gradients (engine.hip, net_forward / net_backward).  share_prefix=0 in the launch plan evaluates
    every row on its own like the reference does: both meet the reference chain, and they agree with each other to
    rounding."""
    g = np.load(f"{GOLDEN}/{fixture}.npz")
    ddim = "ddim" in fixture
    out = {}
    for share in ("1", "0"):
        dm = make_static(100 if ddim else 25, use_apf=ddim, sampler="ddim" if ddim else "ddpm",
                         launch_plan=dict(share_prefix=int(share)))
        if ddim:
            out[share] = step_teacher_forced(dm, g, True)
        else:
            out[share], _ = run(dm, g, 4)
    if ddim:
        print(f"{fixture}: teacher-forced worst step, shared {out['1']:.2e} row-by-row {out['0']:.2e}")
        assert out["1"] < 1e-4 and out["0"] < 1e-4
    else:
        e1, e0 = np.abs(out["1"] - g["chain"]).max(), np.abs(out["0"] - g["chain"]).max()
        print(f"{fixture}: shared {e1:.2e} row-by-row {e0:.2e} between them {np.abs(out['1'] - out['0']).max():.2e}")
        assert e1 < 1e-4 and e0 < 1e-4 and np.abs(out["1"] - out["0"]).max() < 5e-5


def test_graph_replay_is_bitwise_eager_and_repeatable():
    """Captured graph == eager launches, bit for bit, and every job repeats bit for bit from the first one on: a job's first evaluation
    takes its operand scales from the context's canonical calibration (one bf16x6 evaluation on Philox noise of a fixed seed, outside
    any job; ramp_set_calibration_reuse), so no job depends on its predecessors.  With the reuse off every job calibrates itself in its
    first evaluation instead: equally repeatable, a rounding-level different answer."""
    g = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")
    de = make_static(25, use_graph=False)
    a1, _ = run(de, g, 4); a2, _ = run(de, g, 4)
    dm = make_static(25, use_graph=True)
    b1, _ = run(dm, g, 4)
    b2, _ = run(dm, g, 4)          # replays the graph the first job captured
    b3, _ = run(dm, g, 4)
    assert np.array_equal(a1, a2) and np.array_equal(a1, b1) and np.array_equal(b1, b2) and np.array_equal(b2, b3)
    assert np.abs(b2 - g["chain"]).max() < 1e-4
    dm.model.set_calibration_reuse(False)          # every job calibrates itself (bf16x6 first evaluation) and repeats its answer
    c1, _ = run(dm, g, 4); c2, _ = run(dm, g, 4)
    assert np.array_equal(c1, c2) and np.abs(c1 - g["chain"]).max() < 1e-4
    print(f"canonical vs self-calibrating job: {np.abs(c1 - b1).max():.2e}; vs reference {np.abs(b1 - g['chain']).max():.2e} / {np.abs(c1 - g['chain']).max():.2e}")


def test_caller_supplied_sample_fn_is_called_every_step():
    """p_sample_loop honours ``sample_fn`` like the reference's loop (diffusion_model_static.py:232-256, VERDICT r5 missing 2): a
    step function that is not the stock ``ddpm_sample_fn`` is CALLED once per step with the reference's arguments -- t a (B,) long
    tensor on the device, obstacle_pts with the loop's leading axis, forward_t counting up, compose, the **sample_kwargs -- and hard
    conditioning follows every call; ``noise_std_extra_schedule_fn`` receives the 0-d device tensor t[0] (sample_functions.py:24,
    41-44), on the fused path too.  Here the step is the repo's own single-step ddpm_sample_fn behind a counter, so the chain must be
    the reference's (chain_ddpm_plain / extra2: the fixtures the fused job meets)."""
    from ramp_amd.sample_functions import ddpm_sample_fn
    for tag, nwn in (("plain", 0), ("extra2", 2)):
        g = np.load(f"{GOLDEN}/chain_ddpm_{tag}.npz")
        dm = make_static(25)
        calls, sched_args = [], []

        def counted(model, x, hard_conds, context, t, **kw):
            assert model is dm and torch.is_tensor(t) and t.dtype == torch.long and t.is_cuda and t.shape == (4,)
            assert kw["obstacle_pts"].dim() == 4 and kw["obstacle_pts"].shape[0] == 1 and kw["compose"] is False
            calls.append((int(t[0]), kw["forward_t"]))
            return ddpm_sample_fn(model, x, hard_conds, context, t, **kw)

        def sched(t0):
            assert torch.is_tensor(t0) and t0.dim() == 0 and t0.is_cuda      # tensor methods must work, as in the reference
            sched_args.append(int(t0.item()))
            return 0.5

        hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(4, 48).items()}
        with NoiseInjector(list(g["noise"])) as inj:
            chain = dm.run_inference(None, hc, n_samples=4, horizon=48, return_chain=True, obstacle_pts=dev(g["cloud"]),
                                     sample_fn=counted, noise_std_extra_schedule_fn=sched,
                                     n_diffusion_steps_without_noise=nwn).cpu().numpy()
            used = inj.used
        want = list(reversed(range(-nwn, 25)))
        assert [c[0] for c in calls] == want and [c[1] for c in calls] == list(range(25 + nwn)) and sched_args == want
        assert used == g["noise"].shape[0] and chain.shape == g["chain"].shape
        err = np.abs(chain - g["chain"]).reshape(chain.shape[0], -1).max(1)
        print(f"custom sample_fn, {tag}: {len(calls)} calls, final {err[-1]:.2e} max {err.max():.2e}")
        assert err.max() < 1e-4
        # the fused job hands the schedule function the same 0-d device tensors
        sched_args.clear()
        with NoiseInjector(list(g["noise"])):
            fused = dm.run_inference(None, hc, n_samples=4, horizon=48, return_chain=True, obstacle_pts=dev(g["cloud"]),
                                     noise_std_extra_schedule_fn=sched, n_diffusion_steps_without_noise=nwn).cpu().numpy()
        assert sched_args == want and np.abs(fused - g["chain"]).max() < 1e-4
        print(f"   fused job vs step-wise loop: {np.abs(fused - chain).max():.2e}")


def test_a_job_does_not_depend_on_what_ran_before_it():
    """fp16x3 (the default): the same job -- same noise, hard conditions, scene -- gives the same BITS on a fresh context, after a
    different job on the same context, after single evaluations (ramp_score recalibrates its own tables), after a job the range guard
    sent to the bf16x6 kernels, and after a job of another batch size (results used to depend on job history through
    the maxima the previous job's first evaluation had recorded)."""
    g = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")
    other = {"noise": synth.make_noise((26, 4, 48, 4), seed=4321) * 1.7, "cloud": g["cloud"]}
    big = {"noise": synth.make_noise((26, 12, 48, 4), seed=99), "cloud": g["cloud"]}
    fresh, _ = run(make_static(25, use_graph=True), g, 4)
    dm = make_static(25, use_graph=True)
    run(dm, other, 4)
    a, _ = run(dm, g, 4)
    assert np.array_equal(a, fresh)
    x = dev(g["chain"][5]); t = torch.full((4,), 11, dtype=torch.long, device="cuda")
    was = dm.ddim; dm.ddim = True
    for _ in range(3):
        dm.p_mean_variance(x, None, None, t, obstacle_pts=dev(g["cloud"]))          # single evaluations in between (ramp_score)
    dm.ddim = was
    b, _ = run(dm, g, 4)
    assert np.array_equal(b, fresh)
    run(dm, big, 12)                                                                # another batch size: another graph, another canonical table
    c, _ = run(dm, g, 4)
    assert np.array_equal(c, fresh)
    # other start / goal values between the jobs (the canonical calibration used to apply the FIRST job's hard conditions to its
    # input, so a later job with other values inherited scales a fresh context would not have chosen; it now runs on plain Philox noise)
    def run_hc(model, start, goal):
        hcv = {0: torch.tensor(start, dtype=torch.float32), 47: torch.tensor(goal, dtype=torch.float32)}
        with NoiseInjector(list(g["noise"])):
            return model.run_inference(None, hcv, n_samples=4, horizon=48, return_chain=True, obstacle_pts=dev(g["cloud"]),
                                       noise_std_extra_schedule_fn=lambda x: 0.5).cpu().numpy()
    far = ([0.95, -0.9, 0.0, 0.0], [-0.7, 0.85, 0.0, 0.0])
    fresh_far = run_hc(make_static(25, use_graph=True), *far)
    assert np.array_equal(run_hc(dm, *far), fresh_far)                               # after jobs with the default start / goal
    d0, _ = run(dm, g, 4)
    assert np.array_equal(d0, fresh)                                                 # and back
    huge = {"noise": g["noise"].copy(), "cloud": g["cloud"]}
    huge["noise"][4] *= 3e4                                                          # trips the range guard: that job is repeated (fp16x3, the flagged evaluation calibrating)
    with pytest.warns(UserWarning):
        run(dm, huge, 4)
    d, _ = run(dm, g, 4)
    assert np.array_equal(d, fresh)


def step_teacher_forced(dm, g, ddim, noise_scale=0.5, keep=None):
    """Run every loop iteration from the reference's own previous state (teacher forcing): the APF hook is
    discontinuous and stiff, so free-running chains amplify 1e-5 drift (see tests/test_oracle_vs_golden.py)."""
    ref = g["chain"]; n_steps = ref.shape[0] - 1; B = ref.shape[1]
    S_, H_ = dm.state_dim, dm.model.n_support_points
    hc = {k: torch.from_numpy(v).cuda().unsqueeze(0).expand(B, -1) for k, v in synth.default_hard_conds(S_, H_).items()}
    cloud = dev(g["cloud"])
    worst = 0.0
    T = dm.n_diffusion_steps
    if ddim:
        steps = [int(i) for i in dm.ddim_set_timesteps(dm.ddim_num_inference_steps)]
    else:
        steps = [max(i, 0) for i in reversed(range(0, T))]
    for j, t in enumerate(steps):
        if ddim:
            apf = [1 if (dm.APF and j >= dm.apf_ddim["start"]) else 0]
            cfg = dict(dm.apf_ddim) if apf[0] else None
            noise = dev(ref[j])[None]
            x, _ = dm._launch(B, noise, hc, cloud, True, [t], apf, None, cfg, False)
        else:
            apf = [1 if (dm.APF and j > dm.apf_ddpm["after"]) else 0]
            cfg = dict(dm.apf_ddpm, passes=1) if apf[0] else None
            noise = torch.stack([dev(ref[j]), dev(g["noise"][j + 1])])
            x, _ = dm._launch(B, noise, hc, cloud, False, [t], apf, [noise_scale], cfg, False)
        worst = max(worst, float(np.abs(x.cpu().numpy() - ref[j + 1]).max()))
        if keep is not None:
            keep.append(x.cpu().numpy())
    return worst


def assert_as_accurate_as_the_reference(steps, fixture, S, H, T, w, tag, steps_fp32=None):
    """Per step, from the reference's own previous state, against the float64 step (one step amplifies rounding once: not chaotic,
    unlike a free-running w = 5.75 chain, where ANY fp32 evaluation lands 0.5 .. 3 x the reference's distance from the truth depending
    on the rounding realisation -- measured on the H = 64 / T = 50 chain: exact-fp32 MFMA mode 1.9 x, bf16x6 3.1 x, fp16x3 0.9 x).
    Measured (profiles/r05_eps_accuracy.txt): one HIP evaluation is 1.2 .. 2.5 x as far from float64 as a float32 CPU evaluation, in ALL
    three arithmetic modes alike -- the exact-fp32 MFMA mode is the farthest -- i.e. the excess is the single sequential fp32 accumulator
    of an MFMA chain against BLAS's blocked accumulation, not the split-precision operands.  Hence two statements:
      (1) the default fp16x3 step is at most 1.25 x as far from float64 as the exact-fp32 MFMA mode's (the emulation loses nothing);
      (2) it is at most 3 x as far as the reference's own fp32 step, worst step against worst step and on average (20 x inside the
          1e-4 contract: the worst step here is 5e-5 after the sampler's amplification by sqrt_recipm1 = 4.6e3 and w = 5.75)."""
    e_hip, e_ref = util.step_errors_vs_float64(steps, fixture, S, H, T, w)
    msg = (f"{tag}: one step vs float64, worst / mean over steps: HIP {e_hip.max():.2e} / {e_hip.mean():.2e}, reference {e_ref.max():.2e} / "
           f"{e_ref.mean():.2e} (ratios {e_hip.max() / e_ref.max():.2f} / {e_hip.mean() / e_ref.mean():.2f})")
    if steps_fp32 is not None:
        e_f32, _ = util.step_errors_vs_float64(steps_fp32, fixture, S, H, T, w)
        msg += f"; exact-fp32 MFMA mode {e_f32.max():.2e} / {e_f32.mean():.2e}"
    print(msg)
    assert e_hip.max() <= 3.0 * e_ref.max() and e_hip.mean() <= 3.0 * e_ref.mean(), msg
    if steps_fp32 is not None:
        assert e_hip.max() <= 1.25 * e_f32.max() and e_hip.mean() <= 1.25 * e_f32.mean(), msg


def test_ddpm_apf_chain_teacher_forced():
    g = np.load(f"{GOLDEN}/chain_ddpm_apf.npz")
    dm = make_static(25, use_apf=True, use_graph=False)
    worst = step_teacher_forced(dm, g, ddim=False)
    print(f"ddpm apf teacher-forced worst {worst:.2e}")
    assert worst < 1e-4
    # free-running: identical to the plain chain until the hook first fires (forward_t > 20)
    chain, _ = run(make_static(25, use_apf=True), g, 4)
    assert np.abs(chain[:22] - g["chain"][:22]).max() < 1e-4
    assert np.abs(chain[-1] - np.load(f"{GOLDEN}/chain_ddpm_plain.npz")["chain"][-1]).max() > 1e-3


@pytest.mark.parametrize("tag", ["plain", "apf"])
def test_ddim_chain(tag):
    """Script-default mode: DDIM-5 of T=100 (n_diffusion_steps_without_noise is ignored by DDIM)."""
    g = np.load(f"{GOLDEN}/chain_ddim_{tag}.npz")
    dm = make_static(100, use_apf=(tag == "apf"), sampler="ddim")
    if tag == "plain":
        chain, used = run(dm, g, 4, 5)
        assert used == 1 and chain.shape == (6, 4, 48, 4)
        err = np.abs(chain - g["chain"]).reshape(6, -1).max(1)
        print(f"ddim plain: {err}")
        assert err.max() < 1e-4
    else:
        worst = step_teacher_forced(dm, g, ddim=True)
        print(f"ddim apf teacher-forced worst {worst:.2e}")
        assert worst < 1e-4


def test_chain3d_batched_equals_independent_reference_runs():
    """3-D sampler (w = 5.75, DDPM): one batched B=2 call vs two independent n_samples=1 reference runs.

    With w = 5.75 the CFG combine amplifies rounding noise ~12x and the free-running reference chain is only
    reproducible to ~1e-4 by ANY evaluation: a float64 evaluation of the same network differs from the
    reference's fp32 chain by 1.07e-4 (measured, tests/test_oracle_vs_golden.py).  So: (a) every step from the
    reference's own previous state must agree to 1e-4; (b) the free-running chain must sit as close to the
    float64 truth as the reference itself does (within 3x), and within 5e-4 of the reference."""
    from ramp_amd.models import GaussianDiffusionModel3d
    g = np.load(f"{GOLDEN}/chain3d_ddpm.npz")
    u = build_unet(6, 48, True, max_rows=16)
    dm = GaussianDiffusionModel3d(model=u, variance_schedule="exponential", n_diffusion_steps=25,
                                  predict_epsilon=True, use_graph=False).eval().to("cuda")
    steps = []
    worst = step_teacher_forced(dm, g, ddim=False, keep=steps)
    print(f"3d ddpm teacher-forced worst {worst:.2e}")
    assert worst < 1e-4
    u32 = build_unet(6, 48, True, max_rows=16, gemm_mode="fp32")
    d32 = GaussianDiffusionModel3d(model=u32, variance_schedule="exponential", n_diffusion_steps=25, predict_epsilon=True, use_graph=False).eval().to("cuda")
    steps32 = []
    step_teacher_forced(d32, g, ddim=False, keep=steps32)
    assert_as_accurate_as_the_reference(steps, "chain3d_ddpm", 6, 48, 25, 5.75, "3d ddpm", steps32)
    dm.use_graph = True
    chain, used = run(dm, g, 2)
    assert used == 26 and chain.shape == g["chain"].shape
    err = np.abs(chain - g["chain"]).reshape(26, -1).max(1)
    truth = util.oracle64_chain("chain3d_ddpm", 6, 48, 25, 5.75)
    e_ref = np.abs(g["chain"] - truth).max(); e_gpu = np.abs(chain - truth).max()
    print(f"3d ddpm free-running: vs reference {err.max():.2e}; vs float64 truth: reference {e_ref:.2e}, HIP {e_gpu:.2e} (ratio {e_gpu / e_ref:.2f})")
    # free-running: chaotic (see assert_as_accurate_as_the_reference); measured 2.3e-4 from the truth, 2.8e-4 from the reference; bars 2 x
    assert e_gpu < 3 * e_ref and e_gpu < 4.5e-4
    assert err.max() < 5.6e-4
    # the SAME bars on the exact-fp32 MFMA mode, the one mode whose arithmetic is the reference's: if bit-exact fp32
    # products land inside them too, the bars describe the chain, not the fp16x3 emulation
    d32.use_graph = True
    c32, _ = run(d32, g, 2)
    e32 = np.abs(c32 - truth).max(); r32 = np.abs(c32 - g["chain"]).max()
    print(f"   exact-fp32 MFMA mode free-running: vs reference {r32:.2e}; vs float64 truth {e32:.2e} (ratio {e32 / e_ref:.2f})")
    assert e32 < 3 * e_ref and e32 < 4.5e-4 and r32 < 5.6e-4


def make_compose(T, use_apf, sampler=None, use_graph=True, gemm_mode="default"):
    from ramp_amd.models import StaticGaussianDiffusionModel
    u = build_unet(4, 48, False, max_rows=12, gemm_mode=gemm_mode)
    return StaticGaussianDiffusionModel(model=u, n_diffusion_steps=T, predict_epsilon=True, compose=True, use_apf=use_apf,
                                        sampler=sampler, use_graph=use_graph).eval().to("cuda")


def test_compose_static_against_reference_fixture():
    """compose=True (3 rows per trajectory: scene A, scene B, unconditional; e = u + 2(cA-u) + 2(cB-u)) against outputs of
    the reference's own p_mean_variance_compose and compose loops (diffusion_model_static.py:188-229, 259-333):
    the single step, the free-running DDPM chain with use_apf=True (this path has no APF hook in the reference), and the
    DDIM-8 of T=100 + APF chain on the 10-obstacle union cloud, teacher-forced from forward_t = 2 (APF chains are stiff)."""
    g = np.load(f"{GOLDEN}/compose_static.npz")
    clouds = dev(g["clouds"])
    hcn = synth.default_hard_conds(4, 48)
    hc = {k: torch.from_numpy(v) for k, v in hcn.items()}
    # (1) one p_mean_variance_compose
    dm = make_compose(25, False, sampler="ddim")
    t = torch.full((3,), int(g["pmv_t"]), dtype=torch.long, device="cuda")
    mean, _, _, x0, ec = dm.p_mean_variance(dev(g["pmv_x"]), None, None, t, obstacle_pts=clouds, compose=True)
    assert rel(ec.cpu().numpy(), g["pmv_ecomb"]) < 5e-5
    assert np.abs(x0.cpu().numpy() - g["pmv_x0"]).max() < 1e-4 and np.abs(mean.cpu().numpy() - g["pmv_mean"]).max() < 1e-4
    # (2) DDPM T=25, use_apf=True: no hook on the compose path
    dm = make_compose(25, True, sampler="ddpm")
    with NoiseInjector(list(g["ddpm_noise"])) as inj:
        chain = dm.run_inference(None, hc, n_samples=3, horizon=48, return_chain=True, obstacle_pts=clouds,
                                 noise_std_extra_schedule_fn=lambda x: 0.5).cpu().numpy()
        assert inj.used == 26
    err = np.abs(chain - g["ddpm_chain"]).reshape(26, -1).max(1)
    print(f"compose ddpm free-running: final {err[-1]:.2e} max {err.max():.2e}")
    assert err.max() < 2e-4                       # w1 + w2 = 4 amplifies rounding ~2x the CFG chain (oracle32: same bar)
    # (3) DDIM-8 + APF
    dm = make_compose(100, True)
    assert dm.ddim and dm.ddim_num_inference_steps == 8
    with NoiseInjector(list(g["ddim_noise"])) as inj:
        chain = dm.run_inference(None, hc, n_samples=3, horizon=48, return_chain=True, obstacle_pts=clouds).cpu().numpy()
        assert inj.used == 1 and chain.shape == (9, 3, 48, 4)
    assert np.abs(chain[:3] - g["ddim_chain"][:3]).max() < 1e-4          # free-running until the hook first fires
    ref = g["ddim_chain"]
    hcb = {k: torch.from_numpy(v).cuda().unsqueeze(0).expand(3, -1) for k, v in hcn.items()}
    steps = [int(i) for i in dm.ddim_set_timesteps(8)]
    # the APF push is stiff (d direction / d x ~ 1 / distance to the nearest cloud point): at one of these steps even a
    # float64 evaluation of the same step lands 1.6e-4 from the reference's fp32 result.  Bar per step: 1e-4, or as close
    # to the float64 truth as 3x the reference's own distance from it.
    uo = O.UNetOracle(weights(4, 48, False), 4, 48, dtype=np.float64)
    lats = np.stack([uo.encode_scene(g["clouds"][0]), uo.encode_scene(g["clouds"][1])])
    so = O.SamplerOracle(uo, 100, 2.0, dtype=np.float64, sched=dict(np.load(f"{GOLDEN}/schedule_T100.npz")), compose_w=(2.0, 2.0))
    union = np.concatenate([g["clouds"][0], g["clouds"][1][:4]]).reshape(-1, 2)
    truth = so.ddim(g["ddim_noise"][0], hcn, lats, cloud=union, use_apf=True, K=8, teacher=ref)
    worst = 0.0
    for j, tt in enumerate(steps):
        apf = [1 if j >= dm.apf_ddim["start"] else 0]
        x, _ = dm._launch(3, dev(ref[j])[None], hcb, clouds, True, [tt], apf, None, dict(dm.apf_ddim) if apf[0] else None, False)
        got = x.cpu().numpy()
        e_ref = float(np.abs(got - ref[j + 1]).max())
        e_truth, ref_truth = float(np.abs(got - truth[j + 1]).max()), float(np.abs(ref[j + 1] - truth[j + 1]).max())
        worst = max(worst, e_ref)
        assert e_ref < 1e-4 or e_truth < 3 * ref_truth, (j, e_ref, e_truth, ref_truth)
    print(f"compose ddim-8 + apf teacher-forced worst vs reference {worst:.2e}")


def test_compose_3d_against_reference_fixture():
    """3-D compose (w1 = w2 = 5, diffusion_model_3d.py:163-182): one batched B=2 call vs two independent n_samples=1
    reference runs, every step teacher-forced (w1 + w2 = 10 amplifies rounding ~20x per step)."""
    from ramp_amd.models import GaussianDiffusionModel3d
    g = np.load(f"{GOLDEN}/compose_3d.npz")
    u = build_unet(6, 48, True, max_rows=12)
    dm = GaussianDiffusionModel3d(model=u, n_diffusion_steps=25, predict_epsilon=True, compose=True,
                                  use_graph=False).eval().to("cuda")
    assert dm.compose_weights == (5.0, 5.0) and not dm.ddim
    clouds = dev(g["clouds"])
    assert rel(u.encode_scene(clouds).cpu().numpy(), g["latents"]) < 5e-6
    ref = g["chain"]
    hcb = {k: torch.from_numpy(v).cuda().unsqueeze(0).expand(2, -1) for k, v in synth.default_hard_conds(6, 48).items()}
    worst = 0.0
    for j, t in enumerate(reversed(range(25))):
        noise = torch.stack([dev(ref[j]), dev(g["noise"][j + 1])])
        x, _ = dm._launch(2, noise, hcb, clouds, False, [t], [0], [0.5], None, False)
        worst = max(worst, float(np.abs(x.cpu().numpy() - ref[j + 1]).max()))
    print(f"3d compose teacher-forced worst {worst:.2e}")
    assert worst < 1e-4


def test_config5_shape_chain_against_reference_fixture():
    """BASELINE config 5's shape -- 3-D, H = 64, T = 50 DDPM, w = 5.75 -- against a run of the reference
    (diffusion_model_3d.py:185-218): teacher-forced every step at 1e-4; free-running (hipGraph, default fp16x3 mode)
    as close to the float64 truth as the reference's own fp32 chain is (within 3x; T = 50 steps of ~12x rounding amplification)."""
    from ramp_amd.models import GaussianDiffusionModel3d
    g = np.load(f"{GOLDEN}/chain3d_h64_t50.npz")
    u = build_unet(6, 64, True, max_rows=16)
    dm = GaussianDiffusionModel3d(model=u, variance_schedule="exponential", n_diffusion_steps=50,
                                  predict_epsilon=True, use_graph=False).eval().to("cuda")
    steps = []
    worst = step_teacher_forced(dm, g, ddim=False, keep=steps)
    print(f"config-5 shape teacher-forced worst {worst:.2e}")
    assert worst < 1e-4
    u32 = build_unet(6, 64, True, max_rows=16, gemm_mode="fp32")
    d32 = GaussianDiffusionModel3d(model=u32, variance_schedule="exponential", n_diffusion_steps=50, predict_epsilon=True, use_graph=False).eval().to("cuda")
    steps32 = []
    step_teacher_forced(d32, g, ddim=False, keep=steps32)
    assert_as_accurate_as_the_reference(steps, "chain3d_h64_t50", 6, 64, 50, 5.75, "config-5 shape", steps32)
    dm.use_graph = True
    chain, used = run(dm, g, 2)
    assert used == 51 and chain.shape == g["chain"].shape == (51, 2, 64, 6)
    err = np.abs(chain - g["chain"]).max()
    truth = util.oracle64_chain("chain3d_h64_t50", 6, 64, 50, 5.75)
    e_ref = np.abs(g["chain"] - truth).max(); e_gpu = np.abs(chain - truth).max()
    print(f"config-5 shape free-running: vs reference {err:.2e}; vs float64 truth: reference {e_ref:.2e}, HIP {e_gpu:.2e} (ratio {e_gpu / e_ref:.2f})")
    for mode in ("bf16x6", "fp32"):       # the same chain on the other two arithmetic modes: is the distance a property of the mode or of the realisation?
        um = build_unet(6, 64, True, max_rows=16, gemm_mode=mode)
        dmm = GaussianDiffusionModel3d(model=um, variance_schedule="exponential", n_diffusion_steps=50, predict_epsilon=True, use_graph=True).eval().to("cuda")
        cm, _ = run(dmm, g, 2)
        print(f"   {mode}: vs reference {np.abs(cm - g['chain']).max():.2e}, vs float64 truth {np.abs(cm - truth).max():.2e}")
        if mode == "fp32":      # the reference-width mode under the bar that is a property of the chain (3 x the reference's own distance from the
            # truth).  The two absolute bars below are 2 x what the DEFAULT mode measured and do not transfer: on this T = 50
            # chain the exact-fp32 realisation lands 1.9 x the reference's distance (8.2e-4), the fp16x3 one 0.9 x -- the rounding realisation
            # decides, not the width
            assert np.abs(cm - truth).max() < 3 * e_ref and np.abs(cm - g["chain"]).max() < 3 * e_ref + np.abs(g["chain"] - truth).max()
    # free-running: chaotic -- the three arithmetic modes land 0.9 x (fp16x3), 1.9 x (exact fp32 MFMA) and 3.1 x (bf16x6) the
    # reference's own distance from the truth on this chain; bars = 2 x the measured 3.8e-4 from the truth / 5.8e-4 from the reference
    assert e_gpu < 3 * e_ref and e_gpu < 7.6e-4
    assert err < 1.2e-3
    flag = C.c_int32(-1)
    from ramp_amd import _lib as L
    L.check(L.load().ramp_range_status(u.ctx(), C.byref(flag), L.current_stream()))
    assert flag.value == 0


def test_properties_at_scale():
    """Size-independent properties on a larger batch (B=512, chunk capacity 256 rows -> 4 chunks per step):
    determinism, batch-composition independence, hard conditioning exact, states bounded."""
    g = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")
    B = 512
    dm = make_static(25, use_apf=True, max_rows=256)
    noise = synth.make_noise((26, B, 48, 4), seed=99)
    noise[:, :4] = g["noise"]                     # the first four trajectories are the golden ones
    gg = {"noise": noise, "cloud": g["cloud"]}
    a, _ = run(dm, gg, B)          # the context's first job calibrates itself
    b, _ = run(dm, gg, B)          # the following ones continue from their predecessor's calibration ...
    c, _ = run(dm, gg, B)
    assert np.array_equal(b, c)    # ... and repeat bit for bit
    print(f"calibrating vs continuing job: max {np.abs(a - b).max():.2e}")
    assert np.abs(b[:22, :4] - g["chain"][:22]).max() < 1e-4
    assert np.isfinite(a).all() and np.abs(a[-1]).max() <= 1.0 + 0.2     # clamp(x0) + APF push
    plain = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")["chain"]
    assert np.abs(a[:22, :4] - plain[:22]).max() < 1e-4                    # batch neighbours do not matter
    hc = synth.default_hard_conds(4, 48)
    assert np.array_equal(a[-1][:, 0], np.broadcast_to(hc[0], (B, 4))) and np.array_equal(a[-1][:, 47], np.broadcast_to(hc[47], (B, 4)))


def test_dynamic_wrapper_against_reference_fixture():
    """DynamicGaussianDiffusionModel pieces vs the reference: p_mean_variance with the reference's (quirky, Q1) row
    pairing for even and odd B, the per-trajectory static / pursuer APF, sm()."""
    from ramp_amd.apf_dynamic import ObstacleField, avoidance
    from ramp_amd.models import DynamicGaussianDiffusionModel
    g = np.load(f"{GOLDEN}/dynamic_cases.npz")
    u = build_unet(4, 48, False, max_rows=16)
    dm = DynamicGaussianDiffusionModel(model=u, n_diffusion_steps=100, predict_epsilon=True).eval().to("cuda")
    for B in (4, 3):
        x = dev(g[f"pmv{B}/x"]); t = torch.full((B,), 40, dtype=torch.long, device="cuda")
        pts = dev(g["cloud"])[None].repeat(B, 1, 1, 1)
        mean, _, _, x0, ec = dm.p_mean_variance(x, None, None, t, traj_normalized=None, obstacle_pts=pts)
        assert rel(ec.cpu().numpy(), g[f"pmv{B}/ecomb"]) < 5e-5
        assert np.abs(x0.cpu().numpy() - g[f"pmv{B}/x0"]).max() < 1e-4
        assert np.abs(mean.cpu().numpy() - g[f"pmv{B}/mean"]).max() < 1e-4
    dm2 = DynamicGaussianDiffusionModel(model=u, n_diffusion_steps=100, predict_epsilon=True, cfg_mode="intended").eval().to("cuda")
    x = dev(g["pmv4/x"]); t = torch.full((4,), 40, dtype=torch.long, device="cuda")
    ec2 = dm2.p_mean_variance(x, None, None, t, obstacle_pts=dev(g["cloud"]))[4]
    assert rel(ec2.cpu().numpy(), g["pmv4/ecomb"]) > 1e-2          # true CFG differs from the reference's pairing
    thr_s, thr_p, st_s, st_p, w_s, w_p = g["apf/params"]
    field = ObstacleField(static_points=g["apf/static_points"], distance_threshold=thr_s, distance_threshold_pred=thr_p)
    field.set_dynamic_points(g["apf/dynamic_points"])
    tr = dev(g["apf/traj"])
    out_s = avoidance(tr.clone(), field, is_dynamic=False, avoidance_window=int(w_s), avoidance_strength=st_s,
                      avoidance_strength_pred=st_p).cpu().numpy()
    out_d = avoidance(tr.clone(), field, is_dynamic=True, avoidance_window=int(w_p), avoidance_strength=st_s,
                      avoidance_strength_pred=st_p, affected_states=48, goal_state=dev(g["apf/goal"])).cpu().numpy()
    assert np.abs(out_s - g["apf/out_static"]).max() < 5e-7 and np.abs(out_d - g["apf/out_dynamic"]).max() < 5e-7
    one = avoidance(tr[1].clone(), field, is_dynamic=False, avoidance_window=int(w_s), avoidance_strength=st_s).cpu().numpy()
    assert np.array_equal(one, out_s[1])                            # single-trajectory form, like the reference's loop
    assert np.abs(dm.sm(dev(g["sm/s1"]), dev(g["sm/s2"])).cpu().numpy() - g["sm/out"]).max() < 1e-6
    # one full inner step: static APF for all, pursuer pass only for trajectories near the pursuer; finite, goal kept
    xs = dm.ddim_p_sample(tr.clone(), None, None, torch.full((4,), 40, dtype=torch.long, device="cuda"),
                          dev(g["cloud"]), forward_t=3, use_apf=True, use_clipped_model_output=True,
                          obstacle_field=field, pursuer_pos=torch.tensor([0.05, 0.0]))
    assert xs.shape == tr.shape and bool(torch.isfinite(xs).all())


def test_warmup_consumes_one_randn_and_compose_loop_runs():
    """Driver flow of inference_static.py: warmup() (one throw-away score evaluation that consumes one randn,
    diffusion_model_static.py:405-433) then run_inference; and the compose=True DDIM-8 + APF loop end to end."""
    from ramp_amd.models import StaticGaussianDiffusionModel
    g = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")
    dm = make_static(25)
    hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(4, 48).items()}
    extra = synth.make_noise((1, 4, 48, 4), seed=77)
    with NoiseInjector([extra[0]] + list(g["noise"])) as inj:
        dm.warmup(horizon=48, traj_normalized=None, obstacle_pts=dev(g["cloud"]), batch_size=4, device="cuda")
        assert inj.used == 1
        chain = dm.run_inference(None, hc, n_samples=4, horizon=48, return_chain=True, obstacle_pts=dev(g["cloud"]),
                                 noise_std_extra_schedule_fn=lambda x: 0.5).cpu().numpy()
        assert inj.used == 27
    assert np.abs(chain - g["chain"]).max() < 1e-4
    last = dm.run_inference(None, hc, n_samples=4, horizon=48, return_chain=False, obstacle_pts=dev(g["cloud"]),
                            noise_std_extra_schedule_fn=lambda x: 0.5)
    assert last.shape == (4, 48, 4)
    # compose: two scenes + unconditional, DDIM with 8 steps and the APF hook on the 10-obstacle union cloud
    u = build_unet(4, 48, False, max_rows=24)
    dc = StaticGaussianDiffusionModel(model=u, n_diffusion_steps=100, predict_epsilon=True, compose=True,
                                      use_apf=True).eval().to("cuda")
    assert dc.ddim and dc.ddim_num_inference_steps == 8
    clouds = dev(np.stack([synth.make_cloud(6, 64, 2, seed=1), synth.make_cloud(6, 64, 2, seed=2)]))
    out = dc.run_inference(None, hc, n_samples=5, horizon=48, return_chain=True, obstacle_pts=clouds)
    assert out.shape == (9, 5, 48, 4) and bool(torch.isfinite(out).all())
    assert torch.equal(out[-1][:, 0], hc[0].cuda().expand(5, -1)) and torch.equal(out[-1][:, 47], hc[47].cuda().expand(5, -1))


REPLAN_FACTOR = 10.0


def assert_replan_as_accurate_as_the_reference(e64, r64):
    """Every batch the planner ranks, against the FLOAT64 twin of the reference run (replan_chain.npz, cost*/trajs64: the reference planner with
    model, noise and cloud in float64 -- it takes the same discrete decisions), beside the distance r64 of the reference's own fp32 run from it
    (1.3e-5 after the ten plain DDIM steps, 4.3e-5 once the chain has passed through APF pushes and re-selection).  The chain is chaotic:
    replicas of ONE candidate that differ only by the summation order of a wave tile (1e-7 per evaluation) end 2.7e-5 .. 6.6e-5 apart
    (test_dynamic_replanning_reference_run_embedded_in_a_large_batch), and the distance to float64 scatters with the launch plan and the batch
    between 0.25 and 6.7 x r64 (round 5: 1.1e-5 .. 2.9e-4 at the last two batches over four launch plans x three batch sizes; one evaluation is
    equally accurate in all of them, profiles/r05_eps_accuracy.txt).  So the bar is an order of magnitude, not a rounding bound; what pins the planner
    are the discrete decisions asserted beside it -- every selected index, every collision mask, every pursuer update equal to the reference run's."""
    print("   vs float64:", [f"{e:.2e}" for e in e64], "| the reference's fp32 run:", [f"{r:.2e}" for r in r64])
    for j, (e, r) in enumerate(zip(e64, r64)):
        assert e < REPLAN_FACTOR * r, (j, e, r)


@pytest.mark.parametrize("impl", ["graph", "eager-loop", "graph-fp16x3-eager-launch"])
def test_dynamic_replanning_loop_against_reference_run(impl):
    """The receding-horizon planner (diffusion_model_dynamic.py:495-624) against a run of the reference planner with
    the same fake env, numpy seed and torch noise: every batch handed to the cost selection, every selected index,
    every collision mask, every evader state handed to the pursuer dynamics.  'graph': one ramp_sample job for the
    high-level plan + one captured ramp_replan graph per replan (device-side smoothing / APF / costs / selection);
    'eager-loop': the step-at-a-time host loop over the same kernels; the third runs ramp_replan without graph capture."""
    import ramp_amd.cost as cost_mod
    from ramp_amd.models import DynamicGaussianDiffusionModel
    from util import NoiseInjector, StopReplan, make_fake_pursuit_env
    g = np.load(f"{GOLDEN}/replan_chain.npz")
    K = int(g["n_iter"]); B, H, S = g["noise"].shape[1:]
    u = build_unet(4, 48, False, max_rows=16)
    dm = DynamicGaussianDiffusionModel(model=u, n_diffusion_steps=100, predict_epsilon=True,
                                       use_graph=(impl == "graph")).eval().to("cuda")
    log_env, log_cost = [], []
    dataset, sphere = make_fake_pursuit_env(stop_at=K, log=log_env)
    hard = {0: dev(g["hard0"]).repeat(B, 1), H - 1: dev(g["hardN"]).repeat(B, 1)}
    np.random.seed(23)
    if impl == "eager-loop":
        orig = cost_mod.compute_trajectory_costs

        def logged(trajs, pts, **kw):
            out = orig(trajs, pts, **kw)
            log_cost.append((trajs.cpu().numpy().copy(), pts.reshape(-1, 2).shape[0], -1 if out[4] is None else int(out[4]),
                             out[3].cpu().numpy().copy()))
            return out

        cost_mod.compute_trajectory_costs = logged
        try:
            with NoiseInjector(list(g["noise"])):
                with pytest.raises(StopReplan):
                    dm.ddim_p_sample_loop_eager((B, H, S), hard, context={'dataset': dataset}, return_chain=True,
                                                obstacle_pts=dev(g["cloud"]))
        finally:
            cost_mod.compute_trajectory_costs = orig
    else:
        dm.replan_log = []
        with NoiseInjector(list(g["noise"])):
            with pytest.raises(StopReplan):
                dm.ddim_p_sample_loop((B, H, S), hard, context={'dataset': dataset}, return_chain=True,
                                      obstacle_pts=dev(g["cloud"]))
        log_cost = [(e["batch"].cpu().numpy(), e["npts"], e["idx"], e["free"].cpu().numpy()) for e in dm.replan_log]
    assert len(log_cost) == int(g["n_cost"]) and len(log_env) == int(g["n_env"])
    errs, e64, r64 = [], [], []
    for j, (tr, npts, idx, free) in enumerate(log_cost):
        errs.append(float(np.abs(tr - g[f"cost{j}/trajs"]).max()))
        e64.append(float(np.abs(tr - g[f"cost{j}/trajs64"]).max())); r64.append(float(np.abs(g[f"cost{j}/trajs"] - g[f"cost{j}/trajs64"]).max()))
        assert npts == int(g[f"cost{j}/npts"])
        assert idx == int(g[f"cost{j}/idx"]) and np.array_equal(free, g[f"cost{j}/free"]), (j, idx, free)
    for j, (t, st) in enumerate(log_env):
        assert t == int(g[f"env{j}/t"])
        errs.append(float(np.abs(st - g[f"env{j}/state"]).max()))
    print("replan errs", errs)
    assert_replan_as_accurate_as_the_reference(e64, r64)
    assert max(errs[len(log_cost):]) < 1e-5    # the evader states handed to the pursuer dynamics
    assert np.abs(sphere.centers.numpy() - g["pursuer_final"]).max() < 1e-4


@pytest.mark.parametrize("B", [768, 8192])
def test_dynamic_replanning_reference_run_embedded_in_a_large_batch(B):
    """The planner's numerics at the batches the large-launch kernels serve -- B = 768 candidates (73728 tokens at the first
    level) and BASELINE configs[3]'s full B = 8192 (6 x 1365 replicas + 2): the six candidates of the reference run
    (replan_chain.npz, diffusion_model_dynamic.py:495-624) tiled over the batch.  Replicas tie with their originals to rounding,
    so the plan evolves as in the reference run: every ranked batch's first six rows, every selected candidate (the reference's, or a replica
    of it), every collision mask, every pursuer update -- through the captured ramp_replan graphs."""
    from ramp_amd.models import DynamicGaussianDiffusionModel
    from util import NoiseInjector, StopReplan, make_fake_pursuit_env
    g = np.load(f"{GOLDEN}/replan_chain.npz")
    K = int(g["n_iter"]); B0, H, S = g["noise"].shape[1:]
    rep, extra = divmod(B, B0)
    tile = lambda a: np.concatenate([np.tile(a, (rep,) + (1,) * (a.ndim - 1)), a[:extra]])
    u = build_unet(4, 48, False, max_rows=2 * B)
    dm = DynamicGaussianDiffusionModel(model=u, n_diffusion_steps=100, predict_epsilon=True, use_graph=True).eval().to("cuda")
    log_env = []
    dataset, sphere = make_fake_pursuit_env(stop_at=K, log=log_env)
    hard = {0: dev(g["hard0"]).repeat(B, 1), H - 1: dev(g["hardN"]).repeat(B, 1)}
    np.random.seed(23)
    dm.replan_log = []
    with NoiseInjector([tile(n) for n in g["noise"]]):
        with pytest.raises(StopReplan):
            dm.ddim_p_sample_loop((B, H, S), hard, context={'dataset': dataset}, return_chain=True, obstacle_pts=dev(g["cloud"]))
    assert len(dm.replan_log) == int(g["n_cost"]) and len(log_env) == int(g["n_env"])
    assert dm.range_fallbacks == 0
    errs, e64, r64 = [], [], []
    for j, e in enumerate(dm.replan_log):
        tr = e["batch"].cpu().numpy()
        assert tr.shape == (B, H, S)
        errs.append(float(np.abs(tr[:B0] - g[f"cost{j}/trajs"]).max()))
        e64.append(float(np.abs(tr[:B0] - g[f"cost{j}/trajs64"]).max())); r64.append(float(np.abs(g[f"cost{j}/trajs"] - g[f"cost{j}/trajs64"]).max()))
        # replicas stay with their originals: a sample's position inside a wave tile of the fused attention kernels changes its summation
        # order (1e-7 per evaluation), which 15 DDIM steps and the APF push amplify 300- to 600-fold (measured 0.9e-5 in round 4, 2.7e-5 .. 6.6e-5
        # in round 5 depending on the launch plan)
        assert float(np.abs(tr - tile(tr[:B0])).max()) < 1.5e-4
        # rank among the free ones: the winner is the reference's candidate or one of its replicas -- they tie to rounding, and since a wave
        # of the fused attention kernel owns 48 / L samples, a replica at another position in its wave tile sums its keys in another order
        # (1e-7): which of the tied copies has the smallest cost is not determined
        n_free0 = int(g[f"cost{j}/free"].sum())
        ref_idx = int(g[f"cost{j}/idx"])
        assert (e["idx"] == ref_idx) if ref_idx < 0 or n_free0 == 0 else ((e["idx"] - ref_idx) % n_free0 == 0 and e["idx"] >= 0), (j, e["idx"], ref_idx)
        assert np.array_equal(e["free"].cpu().numpy(), tile(g[f"cost{j}/free"])), j
    for j, (t, st) in enumerate(log_env):
        assert t == int(g[f"env{j}/t"])
        errs.append(float(np.abs(st[:B0] - g[f"env{j}/state"]).max()))
    print(f"B = {B} replan errs", errs)
    assert_replan_as_accurate_as_the_reference(e64, r64)
    assert max(errs[len(dm.replan_log):]) < 1e-5
    assert np.abs(sphere.centers.numpy() - g["pursuer_final"]).max() < 1e-4


def test_dynamic_run_inference_terminates_and_respects_constraints():
    """run_inference end to end with the fake env: start near the goal so the loop ends on its own; the returned
    chain / chain_obs / chain_start have the reference's shapes and the executed history is inpainted."""
    from ramp_amd.models import DynamicGaussianDiffusionModel
    from util import make_fake_pursuit_env, FAKE_BOX_CENTRES
    from ramp_amd.apf_dynamic import generate_box_points
    u = build_unet(4, 48, False, max_rows=16)
    dm = DynamicGaussianDiffusionModel(model=u, n_diffusion_steps=100, predict_epsilon=True).eval().to("cuda")
    np.random.seed(3)
    cloud = np.stack([generate_box_points(c, (0.16, 0.16), 64) for c in FAKE_BOX_CENTRES]).astype(np.float32)
    dataset, sphere = make_fake_pursuit_env()
    torch.manual_seed(0)
    hard = {0: torch.tensor([0.05, -0.05, 0.0, 0.0]), 47: torch.tensor([0.15, 0.05, 0.0, 0.0])}
    chain, chain_obs, chain_start = dm.run_inference(context={'dataset': dataset}, hard_conds=hard, n_samples=6,
                                                     return_chain=True, obstacle_pts=dev(cloud), max_iteration=3)
    n_it = len(chain_obs)
    assert 1 <= n_it <= 3
    assert chain.shape == (n_it + 1, 1, 48, 4) and len(chain_start) == n_it + 1
    last = chain[-1, 0].cpu().numpy()
    assert np.allclose(last[0, :2], [0.05, -0.05]) and np.allclose(last[-1], [0.15, 0.05, 0, 0])
    assert np.isfinite(chain.cpu().numpy()).all()


def test_fp16x3_chunking_and_repeat_are_bitwise():
    """fp16x3 operand scales come from the maxima over ALL rows of the previous evaluation (the first one's from the canonical
    calibration, itself an evaluation over all rows), so splitting the rows into chunks (max_rows 16 -> 8 rows x 4 chunks here) must
    not change a bit -- nor must running the job again, with the canonical calibration or with jobs that calibrate themselves."""
    g = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")
    noise = synth.make_noise((26, 16, 48, 4), seed=77)
    noise[:, :4] = g["noise"]
    gg = {"noise": noise, "cloud": g["cloud"]}
    dw = make_static(25, max_rows=64, gemm_mode="fp16x3")
    whole, _ = run(dw, gg, 16); whole2, _ = run(dw, gg, 16)
    dm = make_static(25, max_rows=8, gemm_mode="fp16x3")
    a, _ = run(dm, gg, 16)
    b, _ = run(dm, gg, 16)
    assert np.array_equal(a, whole) and np.array_equal(b, whole2)
    assert np.abs(a[:, :4] - g["chain"]).max() < 1e-4 and np.abs(b[:, :4] - g["chain"]).max() < 1e-4
    assert np.array_equal(a, b)
    dm.model.set_calibration_reuse(False); dw.model.set_calibration_reuse(False)
    c, _ = run(dm, gg, 16); d, _ = run(dm, gg, 16); w3, _ = run(dw, gg, 16)
    assert np.array_equal(c, d) and np.array_equal(c, w3) and np.abs(c[:, :4] - g["chain"]).max() < 1e-4


@pytest.mark.parametrize("how", ["grow", "shrink"])
def test_fp16x3_range_guard_trips_inside_ramp_sample(how):
    """The REAL on-device guard, no hooks: noise that makes an operand leave the range the delayed fp16 scaling assumed.
    'grow': the noise of iteration 3 is 3e4 x larger, so the state (and the residual stream behind the first GroupNorm)
    entering evaluation 4 is 2^14 larger than what evaluation 3 recorded; 'shrink': x_T is 1e4 x larger than every later
    state (posterior_mean_coef2[T-1] = 0 wipes it), so evaluation 0 overflows the canonical scales and evaluation 1 would run on
    maxima 2^13 too large.  The wrapper must discard the job and repeat it.  Round 6: the repeat stays ON THE
    fp16x3 KERNELS -- the guard's state is logged after every evaluation, the first flagged evaluation (ramp_range_trip) and its
    successor run as calibrating ones (bf16x6 kernels recording true maxima) and everything else as before -- a function of the
    job alone: a fresh context repeats the same bits, the
    context's own second run too.  Its chain is as close to the float64 oracle's as the all-bf16x6 job's.  With
    ``fp16_rerun = False`` the repeat is the all-bf16x6 job of rounds 2-5, bitwise; ``fp16_fallback = False`` raises."""
    from ramp_amd import _lib as L
    g0 = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")
    noise = g0["noise"].copy()
    if how == "grow":
        noise[4] *= 3.0e4                 # noise[1 + j] is iteration j's randn_like
    else:
        noise[0] *= 1.0e4
    g = {"noise": noise, "cloud": g0["cloud"]}
    ref, _ = run(make_static(25, gemm_mode="bf16x6"), g, 4)
    assert np.isfinite(ref).all()
    dm = make_static(25, gemm_mode="fp16x3")
    with pytest.warns(UserWarning, match="range guard"):
        fb, _ = run(dm, g, 4)
    ev, site = C.c_int32(-2), C.c_int32(-2)
    L.check(L.load().ramp_range_trip(dm.model.ctx(), C.byref(ev), C.byref(site)))
    assert dm.last_job_mode == "fp16x3-rerun" and dm.range_reruns == 1 and dm.range_fallbacks == 0, (dm.last_job_mode, dm.range_reruns, dm.range_fallbacks)
    # the evaluation whose operands outran what its scales assumed: 'grow' evaluation 4 against evaluation 3's maxima; 'shrink' already
    # evaluation 0, whose x_T is 2^13 above the canonical calibration's -- the undershoot at evaluation 1 is covered by the same repeat
    # (the flagged evaluation AND its successor calibrate)
    assert ev.value == (4 if how == "grow" else 0), ev.value
    assert np.isfinite(fb).all()
    # accuracy: against the float64 oracle on the same noise, next to the all-bf16x6 job's distance (per state, relative to the state's size:
    # the kicked states are 1e4 large)
    from oracle import ramp_oracle as O
    uo = O.UNetOracle(weights(4, 48, False), 4, 48, dtype=np.float64)
    sm = O.SamplerOracle(uo, 25, 2.0, dtype=np.float64, sched=dict(np.load(f"{GOLDEN}/schedule_T25.npz")))
    truth = sm.ddpm(noise[:, :2], synth.default_hard_conds(4, 48), g0["latent"])          # two of the four trajectories (rows are independent)
    size = np.maximum(1.0, np.abs(truth).reshape(26, -1).max(1))
    e_re = np.abs(fb[:, :2] - truth).reshape(26, -1).max(1) / size
    e_x6 = np.abs(ref[:, :2] - truth).reshape(26, -1).max(1) / size
    print(f"guard trip ({how}) at evaluation {ev.value}, site {site.value}: fp16x3 repeat vs float64 {e_re.max():.2e} (all-bf16x6 job {e_x6.max():.2e}); "
          f"repeat vs all-bf16x6 job {(np.abs(fb - ref).reshape(26, -1).max(1) / np.maximum(1.0, np.abs(ref).reshape(26, -1).max(1))).max():.2e}")
    assert e_re.max() < max(1e-4, 3.0 * e_x6.max())
    # a function of the job alone: the same context again (its kept repeat graph), and a fresh context
    with pytest.warns(UserWarning, match="range guard"):
        fb2, _ = run(dm, g, 4)
    with pytest.warns(UserWarning, match="range guard"):
        fb3, _ = run(make_static(25, gemm_mode="fp16x3"), g, 4)
    assert np.array_equal(fb, fb2) and np.array_equal(fb, fb3)
    dm.fp16_rerun = False                                      # rounds 2-5: straight to the bf16x6 kernels
    with pytest.warns(UserWarning, match="bf16x6"):
        old, _ = run(dm, g, 4)
    assert np.array_equal(old, ref) and dm.last_job_mode == "bf16x6" and dm.range_fallbacks == 1
    dm.fp16_rerun = True
    dm.fp16_fallback = False
    with pytest.raises(L.RampHipError, match="fp16 range"):
        run(dm, g, 4)
    dm.fp16_fallback = True
    plain, _ = run(dm, g0, 4)                                  # ordinary noise: no trip, fp16x3 arithmetic again
    flag = C.c_int32(-1)
    L.check(L.load().ramp_range_status(dm.model.ctx(), C.byref(flag), L.current_stream()))
    assert flag.value == 0 and dm.last_job_mode == "fp16x3"
    L.check(L.load().ramp_range_trip(dm.model.ctx(), C.byref(ev), None))
    assert ev.value == -1
    assert np.abs(plain - g0["chain"]).max() < 1e-4
    assert np.array_equal(plain, run(make_static(25, gemm_mode="fp16x3"), g0, 4)[0])        # ... with the bits of a context that never tripped
    assert not np.array_equal(plain, run(make_static(25, gemm_mode="bf16x6"), g0, 4)[0])   # the modes round differently


@pytest.mark.parametrize("gemm_mode", ["fp16x3-tkw", "fp32"])
def test_ddpm_chain_on_outlier_channel_weights(gemm_mode):
    """A CHAIN on trained-transformer statistics (synth.add_outlier_channels: 8 output channels of every attn1.to_out / ff.net.2
    x 2^9): where the delayed, previous-evaluation operand scales of the fp16x3 mode meet operands that move along
    the chain.  Fixture chain_ddpm_outlier.npz = the imported reference's T = 25 run (B = 4) and its float64 twin.  This network is
    violently chaotic -- the reference's OWN fp32 chain is 1e-2 from its float64 twin after three steps and 0.16 at the worst state
    (printed by oracle/make_goldens.py) -- so no evaluation reproduces it free-running; what is asserted: (1) every step FROM THE
    REFERENCE'S previous state, against the float64 step from that state, is as accurate as the reference's own step (<= 3 x worst /
    mean, the bar of the plain chains); (2) the job as a whole runs clean or is repeated -- whichever happens it ends on the fp16x3
    kernels (never a silent degradation, never the all-bf16x6 job) and its first states, before the chaos takes over, meet the
    reference; (3) the exact-fp32 MFMA mode is held to the same bars."""
    from ramp_amd.models import StaticGaussianDiffusionModel, TemporalUnetInference
    from ramp_amd.unet import load_numpy_state_dict
    from oracle import ramp_oracle as O
    g = np.load(f"{GOLDEN}/chain_ddpm_outlier.npz")
    mode, plan = util.split_mode(gemm_mode) if "-" in gemm_mode else (gemm_mode, None)
    sd = synth.add_outlier_channels(weights(4, 48, False))
    u = load_numpy_state_dict(TemporalUnetInference(n_support_points=48, state_dim=4, max_rows=64, gemm_mode=mode, launch_plan=plan), sd)
    dm = StaticGaussianDiffusionModel(model=u.eval().to("cuda"), variance_schedule="exponential", n_diffusion_steps=25, predict_epsilon=True,
                                      compose=False, use_apf=False, sampler="ddpm", use_graph=True).eval().to("cuda")
    # (1) teacher-forced steps against the float64 step of the oracle with the same weights
    keep = []
    import warnings as W
    with W.catch_warnings(record=True) as caught:
        W.simplefilter("always")
        step_teacher_forced(dm, g, ddim=False, keep=keep)
    uo = O.UNetOracle(sd, 4, 48, dtype=np.float64)
    sm = O.SamplerOracle(uo, 25, 2.0, dtype=np.float64, sched=dict(np.load(f"{GOLDEN}/schedule_T25.npz")))
    lat = uo.encode_scene(g["cloud"])
    truth = sm.ddpm(g["noise"], synth.default_hard_conds(4, 48), lat, teacher=g["chain"])
    e_hip = np.array([np.abs(keep[j] - truth[j + 1]).max() for j in range(25)])
    e_ref = np.array([np.abs(g["chain"][j + 1] - truth[j + 1]).max() for j in range(25)])
    trips = [str(w.message) for w in caught if "range guard" in str(w.message)]
    print(f"outlier weights, {gemm_mode}: one step vs float64, worst / mean: HIP {e_hip.max():.2e} / {e_hip.mean():.2e}, reference {e_ref.max():.2e} / "
          f"{e_ref.mean():.2e}; {len(trips)} of 25 single-step jobs tripped the guard; reruns {dm.range_reruns}, bf16x6 fallbacks {dm.range_fallbacks}")
    assert e_hip.max() <= 3.0 * e_ref.max() and e_hip.mean() <= 3.0 * e_ref.mean()
    # (2) the whole job, free-running
    n_re, n_fb = dm.range_reruns, dm.range_fallbacks
    with W.catch_warnings(record=True):
        W.simplefilter("always")
        chain, _ = run(dm, g, 4)
    d = np.abs(chain - g["chain"]).reshape(26, -1).max(1)
    d64 = np.abs(g["chain"] - g["chain64"]).reshape(26, -1).max(1)
    print(f"   free-running: job ended as {dm.last_job_mode} (reruns +{dm.range_reruns - n_re}, bf16x6 fallbacks +{dm.range_fallbacks - n_fb}); "
          "|HIP - reference| per state " + " ".join(f"{v:.1e}" for v in d[:6]) + " ...; the reference's fp32 run vs its float64 twin " + " ".join(f"{v:.1e}" for v in d64[:6]) + " ...")
    assert np.isfinite(chain).all()
    assert dm.last_job_mode in (("fp16x3", "fp16x3-rerun") if mode == "fp16x3" else ("fp32",))
    assert d[1] < 1e-4 and d[2] < max(1e-4, 3.0 * d64[2])


def test_ddpm_chain_at_another_horizon():
    """n_support_points = 40 (levels of 40, 20, 10 and 5 tokens): the reference takes any multiple of 8
    (UnetInference.py:42-56) and so does the engine; free-running 25-step DDPM chain against a run of the reference."""
    from ramp_amd.models import StaticGaussianDiffusionModel
    g = np.load(f"{GOLDEN}/chain_ddpm_h40.npz")
    u = build_unet(4, 40, False, max_rows=64)
    dm = StaticGaussianDiffusionModel(model=u, variance_schedule="exponential", n_diffusion_steps=25, predict_epsilon=True,
                                      compose=False, use_apf=False, sampler="ddpm", use_graph=True).eval().to("cuda")
    chain, used = run(dm, g, g["chain"].shape[1])
    assert used == g["noise"].shape[0] and chain.shape == g["chain"].shape
    err = np.abs(chain - g["chain"]).max()
    print(f"ddpm H=40: max {err:.2e}")
    assert err < 1e-4
    with pytest.raises(Exception):
        build_unet(4, 44, False, max_rows=8).ctx()            # not a multiple of 8: refused at ramp_create, like the reference's shapes


def test_philox_noise_is_host_replicable_and_jobs_draw_it_inside_the_graph():
    """Throughput jobs draw N(0, I) inside the captured job (noise_source='philox': Philox4x32-10 + Box-Muller from a
    (seed, offset) device record) instead of torch.randn on the host stream (sample_functions.py:36).  (1) the stream equals
    its numpy restatement; (2) a job that draws its own noise equals, bit for bit, the same job with that noise INJECTED
    (what the parity runs do), for the first (calibrating) job, the continuing one and a third that replays its graph --
    every job consumes the next block of the stream."""
    from ramp_amd import _lib
    lib = _lib.load()
    n = 4 * 48 * 4 * 26
    out = torch.empty(n + 3, device="cuda")[:n]
    for seed, off in ((0, 0), (0x1234567890ABCDEF, 7), (99, (1 << 33) + 5)):
        _lib.check(lib.ramp_philox_normal(_lib.ptr(out), n, seed, off, None), "ramp_philox_normal")
        z, _ = util.philox_normal(seed, off, n)
        got = out.cpu().numpy()
        assert np.abs(got - z).max() < 2e-5, np.abs(got - z).max()
    big = torch.empty(1 << 22, device="cuda")
    _lib.check(lib.ramp_philox_normal(_lib.ptr(big), big.numel(), 5, 0, None), "ramp_philox_normal")
    assert abs(float(big.mean())) < 2e-3 and abs(float(big.std()) - 1.0) < 2e-3 and float(big.abs().max()) < 6.0
    assert abs(float((big ** 4).mean()) - 3.0) < 3e-2
    # a tail that is not a multiple of four
    tail = torch.full((16,), 7.0, device="cuda")
    _lib.check(lib.ramp_philox_normal(_lib.ptr(tail), 6, 1, 0, None), "ramp_philox_normal")
    assert bool((tail[6:] == 7.0).all()) and np.abs(tail[:6].cpu().numpy() - util.philox_normal(1, 0, 6)[0]).max() < 2e-5

    g = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")
    from ramp_amd.models import StaticGaussianDiffusionModel
    B, H, S, T = 4, 48, 4, 25
    def make(source):
        u = build_unet(S, H, False)
        return StaticGaussianDiffusionModel(model=u, variance_schedule="exponential", n_diffusion_steps=T, predict_epsilon=True,
                                            use_apf=False, sampler="ddpm", use_graph=True, noise_source=source, noise_seed=77).eval().to("cuda")
    dp, dt = make("philox"), make("torch")
    hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(S, H).items()}
    kw = dict(n_samples=B, horizon=H, return_chain=True, traj_normalized=None, obstacle_pts=dev(g["cloud"]), sample_fn=None,
              noise_std_extra_schedule_fn=lambda x: 0.5, n_diffusion_steps_without_noise=0)
    for job in range(3):
        a = dp.run_inference(None, hc, **kw).cpu().numpy()
        seed, off, n_el = dp.last_philox
        assert (seed, n_el) == (77, (T + 1) * B * H * S) and off == job * ((n_el + 3) // 4)
        nz = torch.empty(n_el, device="cuda")
        _lib.check(lib.ramp_philox_normal(_lib.ptr(nz), n_el, seed, off, None), "ramp_philox_normal")
        with NoiseInjector(list(nz.reshape(T + 1, B, H, S))):
            b = dt.run_inference(None, hc, **kw).cpu().numpy()
        assert np.array_equal(a, b), (job, np.abs(a - b).max())
        assert np.isfinite(a).all() and np.abs(a).max() < 10.0


def test_sharded_philox_jobs_reproduce_the_unsharded_job():
    """the sample batch is sharded over GPUs.  With noise_source='philox' the counter is keyed on the GLOBAL
    sample index (ramp_sample_params.philox_sample0 / philox_total; ``set_noise_shard``), so N shards draw exactly what ONE
    job of the same total draws: (1) the x_T states (noise + hard conditioning, no network in between) of two shards equal
    the unsharded job's rows BIT FOR BIT, for consecutive jobs on the same stream; (2) a ragged three-way split likewise;
    (3) ONE score evaluation of a shard equals the unsharded job's rows to 2e-5 (guided combination) -- what differs is the batch the delayed
    power-of-two operand scales were recorded on and a sample's position inside a wave tile, i.e. summation order, not the
    arithmetic contract; (4) over the 25-step chain that rounding-level difference is amplified like any other (x0 = A x - B e
    with B up to 4.6e3 at the first steps): the chains agree to 2e-4, as any two fp32-faithful evaluations of this chain do
    (each sits ~4e-5 from the reference's).  Bitwise sharded == unsharded is NOT claimed."""
    from ramp_amd.models import StaticGaussianDiffusionModel
    g = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")
    H, S, T, B = 48, 4, 25, 48
    hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(S, H).items()}
    cloud = dev(g["cloud"])

    def make(rows):
        u = build_unet(S, H, False, max_rows=rows)
        return StaticGaussianDiffusionModel(model=u, n_diffusion_steps=T, predict_epsilon=True, sampler="ddpm", use_graph=True,
                                            noise_source="philox", noise_seed=4242).eval().to("cuda")
    kw = dict(horizon=H, return_chain=True, obstacle_pts=cloud, noise_std_extra_schedule_fn=lambda x: 0.5)
    whole = make(2 * B)
    parts = [make(2 * 24), make(2 * 24)]
    rag = [make(2 * 20), make(2 * 20), make(2 * 20)]
    for job in range(2):
        ref = whole.run_inference(None, hc, n_samples=B, **kw)
        got = []
        for r, dm in enumerate(parts):
            dm.set_noise_shard(24 * r, B)
            got.append(dm.run_inference(None, hc, n_samples=24, **kw))
        got = torch.cat(got, dim=1)
        assert torch.equal(got[0], ref[0]), job                  # x_T: the same elements of the same stream
        d = float((got - ref).abs().max())
        print(f"job {job}: 2 shards vs 1 job, all {T + 1} states: max {d:.2e}")
        assert d < 2e-4                                          # two fp32-faithful evaluations of a 25-step chain (each is ~4e-5 from the reference)
        cuts = [(0, 20), (20, 36), (36, 48)]
        got = []
        for (a, b), dm in zip(cuts, rag):
            dm.set_noise_shard(a, B)
            got.append(dm.run_inference(None, hc, n_samples=b - a, **kw))
        got = torch.cat(got, dim=1)
        assert torch.equal(got[0], ref[0]), job
        assert float((got - ref).abs().max()) < 2e-4
        assert whole.last_philox == parts[0].last_philox == rag[2].last_philox      # every shard advanced the stream alike
    with pytest.raises(ValueError):
        parts[0].set_noise_shard(40, B)
        parts[0].run_inference(None, hc, n_samples=24, **kw)
    # one evaluation: the shard's rows against the same rows inside the whole batch
    x = ref[7]
    tt = lambda n: torch.full((n,), 17, dtype=torch.long, device="cuda")
    was = whole.ddim
    outs = []
    for dm, xs in ((whole, x), (parts[1], x[24:].contiguous())):
        dm.ddim = True
        for _ in range(2):        # (the first p_mean_variance after a job recalibrates: compare steady fp16x3 evaluations)
            _, _, _, _, ec = dm.p_mean_variance(xs, None, None, tt(xs.shape[0]), obstacle_pts=cloud)
        dm.ddim = was
        outs.append(ec)
    e1 = rel(outs[1].cpu().numpy(), outs[0][24:].cpu().numpy())
    print(f"one evaluation, shard rows vs the same rows of the whole batch: {e1:.2e}")
    assert e1 < 2e-5          # (e_comb = 3 c - 2 u: five times the ~2.5e-6 by which two fp32-faithful evaluations of eps differ)


def test_sharded_philox_jobs_at_config5_shape():
    """The numbers the first 8-GPU run of BASELINE configs[4] will be read against: the sharded == unsharded check at
    config 5's SHAPE -- 3-D, H = 64, T = 50 DDPM, w = 5.75, where the chain is chaotic -- B = 8 as one job and as two shards of 4 with the
    Philox stream keyed on the global sample index: (1) x_T bit for bit; (2) every step of a shard FROM THE UNSHARDED JOB'S previous state
    (teacher-forced, the job's own noise regenerated with ramp_philox_normal) within 1e-4 of the unsharded job's next state; (3) the
    free-running chains: printed per state and bounded by 2 x the measured distance -- two fp32-faithful evaluations of this chain part
    ways like the arithmetic modes do (test_config5_shape_chain_against_reference_fixture: 4e-4 .. 1.3e-3 from the float64 truth)."""
    from ramp_amd import _lib
    from ramp_amd.models import GaussianDiffusionModel3d
    g = np.load(f"{GOLDEN}/chain3d_h64_t50.npz")
    H, S, T, B = 64, 6, 50, 8
    hc = {k: torch.from_numpy(v) for k, v in synth.default_hard_conds(S, H).items()}
    cloud = dev(g["cloud"])

    def make(rows):
        u = build_unet(S, H, True, max_rows=rows)
        return GaussianDiffusionModel3d(model=u, n_diffusion_steps=T, predict_epsilon=True, use_graph=True, noise_source="philox",
                                        noise_seed=777).eval().to("cuda")
    kw = dict(horizon=H, return_chain=True, obstacle_pts=cloud, noise_std_extra_schedule_fn=lambda x: 0.5)
    whole = make(2 * B)
    parts = [make(2 * 4), make(2 * 4)]
    ref = whole.run_inference(None, hc, n_samples=B, **kw)
    seed, off, n_el = whole.last_philox
    got = []
    for r, dm in enumerate(parts):
        dm.set_noise_shard(4 * r, B)
        got.append(dm.run_inference(None, hc, n_samples=4, **kw))
        assert dm.last_philox == whole.last_philox
    got = torch.cat(got, dim=1)
    assert torch.equal(got[0], ref[0])                           # x_T: the same elements of the same stream
    d = (got - ref).abs().reshape(T + 1, -1).max(1).values.cpu().numpy()
    print(f"config-5 shape, 2 shards vs 1 job, free-running per state: first 1e-5 at state {int(np.argmax(d > 1e-5))}, max {d.max():.2e} (final {d[-1]:.2e})")
    # (2) teacher-forced: the job's noise block, as the job drew it
    nz = torch.empty(n_el, device="cuda")
    _lib.check(_lib.load().ramp_philox_normal(_lib.ptr(nz), n_el, seed, off, None), "ramp_philox_normal")
    nz = nz.reshape(T + 1, B, H, S)
    hcb = {k: v.cuda().unsqueeze(0).expand(4, -1) for k, v in hc.items()}
    tf = []
    for j in range(T):
        for r, dm in enumerate(parts):
            rows = slice(4 * r, 4 * r + 4)
            x, _ = dm._launch(4, torch.stack([ref[j, rows].contiguous(), nz[j + 1, rows].contiguous()]), hcb, cloud, False, [T - 1 - j], [0], [0.5], None, False)
            tf.append(float((x - ref[j + 1, rows]).abs().max()))
    print(f"   teacher-forced from the unsharded job's states: worst step {max(tf):.2e}, mean {np.mean(tf):.2e}")
    assert max(tf) < 1e-4
    # (3) free-running: 2 x the measured 5.5e-4 (round-6 library; the source is the chain, see the docstring)
    assert d.max() < 1.1e-3 and bool(torch.isfinite(got).all())


def test_predict_epsilon_false_and_the_public_helpers():
    """The reference constructor's default predict_epsilon=False (the combined network output IS x0,
    diffusion_model_static.py:28, 109-118) and the sampler class's public helpers, against outputs of the imported reference
    (tests/golden/boundary_cases.npz): one p_mean_variance, a free-running T = 25 DDPM chain, one static ddim_p_sample step
    with and without the APF hook, p_mean_variance_compose through its own name."""
    from ramp_amd.models import StaticGaussianDiffusionModel
    g = np.load(f"{GOLDEN}/boundary_cases.npz")
    H, S = 48, 4
    cloud = dev(g["cloud"])
    t = torch.full((3,), int(g["t"]), dtype=torch.long, device="cuda")
    for pe, tag in ((True, "eps"), (False, "x0")):
        u = build_unet(S, H, False, max_rows=16)
        dm = StaticGaussianDiffusionModel(model=u, n_diffusion_steps=25, predict_epsilon=pe, sampler="ddim").eval().to("cuda")
        mean, _, _, x0, ec = dm.p_mean_variance(dev(g["x"]), None, None, t, obstacle_pts=cloud.unsqueeze(0))
        assert rel(ec.cpu().numpy(), g[f"pmv_ecomb_{tag}"]) < 5e-5
        assert np.abs(x0.cpu().numpy() - g[f"pmv_x0_{tag}"]).max() < 1e-4, tag
        assert np.abs(mean.cpu().numpy() - g[f"pmv_mean_{tag}"]).max() < 1e-4, tag
    # the chain: x0-prediction, DDPM, free-running, graph
    u = build_unet(S, H, False, max_rows=16)
    dm = StaticGaussianDiffusionModel(model=u, n_diffusion_steps=25, sampler="ddpm", use_graph=True).eval().to("cuda")   # predict_epsilon omitted: False
    assert dm.predict_epsilon is False
    # (with the network output taken as x0 the synthetic-weight chain doubles a perturbation every step -- 1.7e-6 after the first step, O(1) after
    # 25, in every arithmetic mode alike -- so free-running it is compared over its first six states, and over all 25 steps from the reference's
    # own previous state)
    chain, used = run(dm, {"noise": g["x0_noise"], "cloud": g["cloud"]}, 2)
    err = np.abs(chain - g["x0_chain"]).reshape(chain.shape[0], -1).max(1)
    worst = step_teacher_forced(dm, {"chain": g["x0_chain"], "noise": g["x0_noise"], "cloud": g["cloud"]}, ddim=False)
    print(f"predict_epsilon=False DDPM chain: free-running {' '.join(f'{e:.1e}' for e in err[:8])} ...; per step from the reference's state {worst:.2e}")
    assert used == 26 and chain.shape == g["x0_chain"].shape and np.isfinite(chain).all()
    assert err[:6].max() < 5e-5 and worst < 5e-5
    # one static DDIM step, T = 100 / K = 5, t = 40, forward_t = 2
    hc = {k: torch.from_numpy(v).cuda()[None].repeat(3, 1) for k, v in synth.default_hard_conds(S, H).items()}
    for apf, key in ((False, "ddim_out"), (True, "ddim_out_apf")):
        u = build_unet(S, H, False, max_rows=16)
        dm = StaticGaussianDiffusionModel(model=u, n_diffusion_steps=100, predict_epsilon=True, use_apf=apf).eval().to("cuda")
        out = dm.ddim_p_sample(dev(g["ddim_x"]), hc, None, torch.full((3,), 40, dtype=torch.long, device="cuda"), cloud.unsqueeze(0),
                               forward_t=2, eta=0.0, use_clipped_model_output=True)
        e = np.abs(out.cpu().numpy() - g[key]).max()
        print(f"static ddim_p_sample (apf={apf}): {e:.2e}")
        assert e < 1e-4, key
    # compose through the reference's method name
    gc_ = np.load(f"{GOLDEN}/compose_static.npz")
    dmc = make_compose(25, False, sampler="ddim")
    tt = torch.full((3,), int(gc_["pmv_t"]), dtype=torch.long, device="cuda")
    mean, _, _, x0, ec = dmc.p_mean_variance_compose(dev(gc_["pmv_x"]), None, None, tt, obstacle_pts=dev(gc_["clouds"]), compose=True)
    assert rel(ec.cpu().numpy(), gc_["pmv_ecomb"]) < 5e-5 and np.abs(mean.cpu().numpy() - gc_["pmv_mean"]).max() < 1e-4
    with pytest.raises(ValueError):
        dm.p_mean_variance_compose(dev(g["x"]), None, None, t, obstacle_pts=cloud)
