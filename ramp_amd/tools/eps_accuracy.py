"""How accurate is ONE score evaluation, mode by mode?  eps of the HIP path in every arithmetic mode / launch plan against the float64
oracle, beside the error of a float32 CPU evaluation of the same network (the numpy oracle in float32 and the eager PyTorch-CPU model:
two independent fp32 evaluations, the yardstick for "as accurate as the reference").  Inputs: states of the 3-D reference chain
(w = 5.75) at the steps where the sampler amplifies eps most.  eps_accuracy.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ramp_oracle as O
import util
from util import GOLDEN, build_unet, dev, weights

g = np.load(f"{GOLDEN}/chain3d_ddpm.npz")
S, H = 6, 48
u64 = O.UNetOracle(weights(S, H, True), S, H, obstacle_3d=True, dtype=np.float64)
u32 = O.UNetOracle(weights(S, H, True), S, H, obstacle_3d=True, dtype=np.float32)
lat = g["latent"]
for j in (1, 2, 5, 12, 24):
    t = 24 - j
    x = g["chain"][j]                                   # (2, H, S)
    x2 = np.repeat(x, 2, axis=0); tt = np.full((4,), t); lats = np.tile(lat[None], (4, 1)); lats[1::2] = 0
    truth = u64.score(x2.astype(np.float64), tt, lats.astype(np.float64))
    e32 = u32.score(x2, tt, lats.astype(np.float32))
    sc = np.abs(truth).max()
    line = f"step {j} (t={t}): |eps| {sc:.3f}; numpy-fp32 {np.abs(e32 - truth).max() / sc:.2e}"
    for mode in ("fp32", "bf16x6", "fp16x3", "fp16x3-tkw"):
        base, plan = util.split_mode(mode)
        m = build_unet(S, H, True, max_rows=8, gemm_mode=base, launch_plan=plan)
        xd = dev(x2); td = torch.from_numpy(tt).cuda()
        m.set_scene(torch.cat([dev(lat)[None], torch.zeros(1, lat.shape[0], device="cuda")]), [0, 1])
        m.prepare_time_table(25)
        from ramp_amd import _lib
        eps = torch.empty((4, H, S), device="cuda")
        for _ in range(3):
            _lib.check(_lib.load().ramp_score(m.ctx(), _lib.ptr(dev(x)), 2, 2, int(t), None, _lib.ptr(eps), _lib.current_stream()))
        line += f" | {mode} {np.abs(eps.cpu().numpy() - truth).max() / sc:.2e}"
        m._destroy_ctx()
    print(line, flush=True)
