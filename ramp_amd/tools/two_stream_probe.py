"""Does ONE GPU sample faster when the job's batch is split over TWO contexts whose graphs replay concurrently on two streams?
Every kernel of the library launches one persistent block per CU; a launch whose tile count is not a multiple of the CU count leaves CUs idle in
its last round (the L = 6 level's feed-forward: 384 tiles on 256 CUs), and consecutive launches of one stream cannot overlap.  Two half-batch
jobs in flight let the dispatcher fill one stream's tails and gaps with the other's blocks.  The halves draw the noise of the unsplit job
(Philox keyed on the global sample index, set_noise_shard), exactly like two ranks of a sharded job.

usage (GPU box): python ramp_amd/tools/two_stream_probe.py [B] [rounds] [n_streams]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from ramp_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bench.WL = bench.WORKLOADS[2]
bench.MAX_ROWS = None
bench.NOISE = "philox"
WL = bench.WL
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cloud = torch.from_numpy(synth.make_cloud(WL["cloud"][0], WL["cloud"][1], 2, seed=42)).to(dev)
hc = {k: torch.from_numpy(v).to(dev) for k, v in synth.default_hard_conds(WL["S"], WL["H"]).items()}


def job(dm, n):
    return bench.run_job(dm, n, cloud, hc, 1)


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


# ---- the unsplit job
dm, _ = bench.build_model(B, dev)
dm.noise_seed = 1234
job(dm, B); job(dm, B)
whole = []
# ---- the same batch as NS shards, one context + stream + host thread each
parts = []
per = B // NS
for i in range(NS):
    d, _ = bench.build_model(per, dev)
    d.noise_seed = 1234
    d.set_noise_shard(i * per, B)
    parts.append((d, torch.cuda.Stream(device=dev)))
for d, s in parts:                                     # graph capture + canonical calibration, one context at a time
    with torch.cuda.stream(s):
        job(d, per); job(d, per)
    torch.cuda.synchronize()


def split_job():
    outs = [None] * NS
    errs = []

    def run(i):
        try:
            d, s = parts[i]
            torch.cuda.set_device(dev)
            with torch.cuda.stream(s):
                outs[i] = job(d, per)
                s.synchronize()
        except BaseException as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(i,)) for i in range(NS)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    return torch.cat(outs, 0)


for r in range(ROUNDS):
    tw, ow = timed(lambda: job(dm, B), 2)
    ts, os_ = timed(split_job, 2)
    d = float((ow - os_).abs().max())
    print(f"round {r}: one job of {B}: {tw * 1e3:8.1f} ms = {B / tw:7.1f} traj/s | {NS} concurrent jobs of {per}: {ts * 1e3:8.1f} ms = {B / ts:7.1f} traj/s "
          f"({(tw / ts - 1) * 100:+.1f} %) | max |unsplit - split| = {d:.2e}; flags {getattr(dm, 'range_fallbacks', 0)} "
          f"{[getattr(p[0], 'range_fallbacks', 0) for p in parts]}", flush=True)
# the halves one after the other on one stream (what splitting alone costs)
def serial_job():
    return torch.cat([job(d, per) for d, _ in parts], 0)
tq, _ = timed(serial_job, 2)
print(f"the {NS} shards one after the other: {tq * 1e3:8.1f} ms = {B / tq:7.1f} traj/s")
