"""world_size-2 sharding / all-gather on CPU with the gloo backend (the GPU box runs the same code on RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ramp_amd import dist as rdist


def test_shard_counts():
    assert rdist.shard_counts(10, 4) == [3, 3, 2, 2]
    assert rdist.shard_counts(8, 8) == [1] * 8
    assert rdist.shard_counts(3, 4) == [1, 1, 1, 0]
    for n, w in ((4096, 8), (65536, 8), (35, 2), (1, 2)):
        c = rdist.shard_counts(n, w)
        assert sum(c) == n and max(c) - min(c) <= 1
        spans = [rdist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    rdist.init_process_group("gloo")
    truth = torch.arange(n_total * 48 * 4, dtype=torch.float32).reshape(n_total, 48, 4)

    def sample_local(start, stop):      # stands in for the per-GPU sampler: rows are a function of the global index
        return truth[start:stop].clone() * 2.0

    out = rdist.sample_sharded(sample_local, n_total)
    ok = bool(torch.equal(out, truth * 2.0))
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ok, float(t)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_sharded_sampling_allgather_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and all(r[2] == 2.0 for r in res)


def test_launch_local_ranks(tmp_path):
    """The launcher behind `python bench.py --gpus N` (no torchrun): N children with the torchrun environment, rank 0's
    line relayed, non-zero exit when a rank fails (the others are terminated instead of hanging in a collective)."""
    import json
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rank_probe.py")
    code = ("import sys; sys.path.insert(0, %r); from ramp_amd import dist as d; "
            "sys.exit(d.launch_local_ranks([%r] + sys.argv[1:], 2, timeout=240))"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), probe))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    ok = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    line = json.loads(ok.stdout.strip().splitlines()[-1])
    assert line == {"n_gpus": 2, "max": 2.0, "gathered": [0.0, 0.0, 1.0, 1.0]}
    bad = subprocess.run([sys.executable, "-c", code, "fail"], capture_output=True, text=True, env=env, timeout=300)
    assert bad.returncode != 0
