"""Shared helpers for the test-suite."""
import os

import numpy as np
import torch

from ramp_amd import synth
from ramp_amd.spec import make_unet_spec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_SD = {}


def weights(S, H, o3):
    key = (S, o3)
    if key not in _SD:
        _SD[key] = synth.make_unet_state_dict(make_unet_spec(S, H, obstacle_3d=o3))
    return _SD[key]


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


PLANS = {   # launch plans the parity tests force onto the small fixtures (ramp_launch_plan; row thresholds: 0 never, 1 always)
    "": dict(ff_fused_rows=0, ffx_rows=0, tkl_rows=0, atk_rows=0, tkc_rows=0, tkw_rows=0),
    "fusedff": dict(ff_fused_rows=1, ffx_rows=0, tkl_rows=0, atk_rows=0, tkc_rows=0, tkw_rows=0),   # FF1 -> GEGLU -> FF2 forward in one launch (gemm.hip, ff_fwd_kernel)
    "ffx": dict(ff_fused_rows=0, ffx_rows=1, tkl_rows=0, atk_rows=0, tkc_rows=0, tkw_rows=0),       # token-owning fused feed-forward, forward and backward (ffx.hip)
    "tok": dict(ff_fused_rows=0, ffx_rows=1, tkl_rows=1, atk_rows=0, tkc_rows=0, tkw_rows=0),       # + token-owning LN1 -> QKV, out-projection, d(o) (tkl.hip): round 3's bench plan
    "atk": dict(ff_fused_rows=0, ffx_rows=1, tkl_rows=1, atk_rows=1, tkc_rows=0, tkw_rows=0),       # + self-attention fused with the out-projection (atk.hip)
    "tkc": dict(ff_fused_rows=0, ffx_rows=1, tkl_rows=1, atk_rows=1, tkc_rows=1, tkw_rows=0),       # + the narrow k = 5 convolutions on sample-owning waves (tkc.hip): round 4's bench plan
    "tkw": dict(ff_fused_rows=0, ffx_rows=1, tkl_rows=1, atk_rows=1, tkc_rows=1, tkw_rows=1),       # + the wide k = 5 convolutions with GroupNorm + Mish fused around them (tkw.hip): the bench's plan
    # round 6: the fused feed-forward of every plan above runs on v_mfma_f32_16x16x32_f16 (ffx16.hip, mfma16 = 1 by default); this one keeps
    # the 32x32x16 pair of ffx.hip under the same fixtures
    "m32": dict(ff_fused_rows=0, ffx_rows=1, tkl_rows=1, atk_rows=1, tkc_rows=1, tkw_rows=1, mfma16=0),
}


def split_mode(mode):
    """'fp16x3-ffx' -> ('fp16x3', launch plan)"""
    base, _, plan = mode.partition("-")
    return base, PLANS[plan]


def build_unet(S, H, o3, max_rows=64, debug=False, gemm_mode="default", launch_plan=None):
    from ramp_amd.models import TemporalUnetInference
    from ramp_amd.unet import load_numpy_state_dict
    m = TemporalUnetInference(n_support_points=H, state_dim=S, obstacle_3d=o3, max_rows=max_rows, debug_taps=debug,
                              gemm_mode=gemm_mode, launch_plan=launch_plan)
    load_numpy_state_dict(m, weights(S, H, o3))
    return m.eval().to("cuda")


class NoiseInjector:
    """Replace torch.randn / randn_like by a pre-generated list (same trick the golden generator used on
    the reference), so the product's own loops consume exactly the golden noise."""

    def __init__(self, arrays, device="cuda"):
        self.q = [a.to(device) if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a)).to(device)
                  for a in arrays]
        self.used = 0

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like

        def randn(*shape, **kw):
            if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
                shape = tuple(shape[0])
            t = self.q[self.used]; self.used += 1
            assert tuple(t.shape) == tuple(shape), (t.shape, shape)
            return t.clone()

        def randn_like(x, **kw):
            t = self.q[self.used]; self.used += 1
            assert t.shape == x.shape
            return t.clone()

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


FAKE_BOX_CENTRES = [[-0.3, 0.2], [0.35, -0.25], [0.1, 0.55], [-0.5, -0.5], [0.7, -0.7], [-0.7, 0.7]]


class StopReplan(Exception):
    pass


def make_fake_pursuit_env(stop_at=None, log=None):
    """The same stand-in for context['dataset'].env that oracle/make_goldens.py drove the reference planner with:
    boxes with .centers / .sizes, a one-sphere pursuer with .centers (1,2) / .radii (1,) and a deterministic
    update_centers(t, current_state) stepping 0.05 toward the mean evader position."""
    from types import SimpleNamespace as NS

    class Sphere:
        def __init__(self):
            self.centers = torch.tensor([[0.6, 0.55]], dtype=torch.float32)
            self.radii = torch.tensor([0.1], dtype=torch.float32)

        def update_centers(self, t, current_state):
            if log is not None:
                log.append((int(t), current_state.detach().cpu().numpy().copy()))
            if stop_at is not None and t >= stop_at:
                raise StopReplan()
            tgt = current_state.detach().cpu().float().mean(dim=0)[:2]
            d = tgt - self.centers[0]
            n = float(torch.linalg.norm(d))
            step = d * (0.05 / n) if n > 0.05 else d
            self.centers = (self.centers[0] + step).unsqueeze(0)

    boxes = NS(centers=torch.tensor(FAKE_BOX_CENTRES), sizes=torch.full((6, 2), 0.16))
    sphere = Sphere()
    env = NS(obj_fixed_list=[NS(fields=[boxes])], obj_extra_list=[NS(fields=[sphere])])
    return NS(env=env), sphere


_TRUTH = {}


def oracle64_chain(fixture, S, H, T, w, o3=True, teacher=False):
    """The float64 oracle's DDPM chain on a chain fixture's inputs (its noise, its scene latent): the truth that both the
    reference's fp32 chain and the HIP chain are measured against where CFG (w = 5.75) amplifies rounding ~12x per step.
    teacher=False: free-running.  teacher=True: state j + 1 is the float64 step FROM THE REFERENCE'S OWN state j (teacher
    forcing), so |reference[j + 1] - truth[j + 1]| is the reference's rounding error of ONE step -- the yardstick for "as
    accurate as the reference" that a chaotic free-running chain cannot give.  Cached per process (H = 64 / T = 50: ~20 s)."""
    key = (fixture, S, H, T, w, teacher)
    if key not in _TRUTH:
        from oracle import ramp_oracle as O
        g = np.load(f"{GOLDEN}/{fixture}.npz")
        uo = O.UNetOracle(weights(S, H, o3), S, H, obstacle_3d=o3, dtype=np.float64)
        sm = O.SamplerOracle(uo, T, w, dtype=np.float64, sched=dict(np.load(f"{GOLDEN}/schedule_T{T}.npz")))
        _TRUTH[key] = sm.ddpm(g["noise"], synth.default_hard_conds(S, H), g["latent"], teacher=g["chain"] if teacher else None)
    return _TRUTH[key]


def step_errors_vs_float64(steps_hip, fixture, S, H, T, w):
    """(HIP, reference) per-step distances from the float64 step taken from the reference's own previous state.
    steps_hip[j] = the HIP step from reference state j (what the teacher-forced tests compute)."""
    g = np.load(f"{GOLDEN}/{fixture}.npz")
    truth = oracle64_chain(fixture, S, H, T, w, teacher=True)
    n = g["chain"].shape[0] - 1
    e_hip = np.array([np.abs(steps_hip[j] - truth[j + 1]).max() for j in range(n)])
    e_ref = np.array([np.abs(g["chain"][j + 1] - truth[j + 1]).max() for j in range(n)])
    return e_hip, e_ref


def philox_normal(seed, offset, n):
    """Host replica of ramp_philox_normal (include/ramp_hip.h; sampler.hip philox_normal_kernel): returns (z float32 [n],
    r uint32 [4 * ceil(n / 4)]) -- Philox4x32-10 words (exact) and their Box-Muller normals (float32 numpy math, so within a
    few ulp of the device's logf / sincosf)."""
    ng = (n + 3) // 4
    ctr = (np.arange(ng, dtype=np.uint64) + np.uint64(offset))
    c = [(ctr & np.uint64(0xFFFFFFFF)).astype(np.uint64), (ctr >> np.uint64(32)).astype(np.uint64),
         np.zeros(ng, np.uint64), np.zeros(ng, np.uint64)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]; p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & M, p0 & M]
        k0 = (k0 + np.uint64(0x9E3779B9)) & M; k1 = (k1 + np.uint64(0xBB67AE85)) & M
    r = np.stack(c, axis=1).astype(np.uint32)                       # (ng, 4)
    u = ((r >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    z = np.empty((ng, 4), np.float32)
    for h in range(2):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[:, 2 * h])).astype(np.float32)
        ang = (np.float32(6.283185307179586) * u[:, 2 * h + 1]).astype(np.float32)
        z[:, 2 * h] = rad * np.cos(ang); z[:, 2 * h + 1] = rad * np.sin(ang)
    return z.reshape(-1)[:n], r.reshape(-1)
