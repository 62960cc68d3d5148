"""Determinism of repeated sampling jobs at a multi-chunk size (B = 512 trajectories, 256-row chunks, APF on): job 1 calibrates,
jobs 2.. continue from their predecessor's calibration and must repeat bit for bit (max |difference| to the previous job)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ramp_amd import synth
from util import GOLDEN
import test_gpu_sampler as T
g = np.load(f"{GOLDEN}/chain_ddpm_plain.npz")
B = 512
dm = T.make_static(25, use_apf=True, max_rows=256)
noise = synth.make_noise((26, B, 48, 4), seed=99); noise[:, :4] = g["noise"]
gg = {"noise": noise, "cloud": g["cloud"]}
runs = [T.run(dm, gg, B)[0] for _ in range(5)]
for i in range(1, 5):
    d = np.abs(runs[i] - runs[i - 1])
    print(i, "max diff vs previous", d.max(), "first differing state", (d.reshape(26, -1).max(1) > 0).argmax() if d.max() > 0 else -1, flush=True)
