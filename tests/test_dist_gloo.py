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


# ---- multi-GPU proofs that need no GPU: verified gather, merged selection, bench.py's --gpus decision ---------------------
def _costs_select(mask, plen, smooth, w_smooth, w_len):
    """compute_trajectory_costs' selection (cost.py:56-88) on per-candidate scalars, float32 like torch: the stand-in for the
    HIP selection kernel in these CPU tests (same contract as ramp_select_from_costs)."""
    free = (mask.numpy() == 0)
    if not free.any():
        return 0, -1, -1
    pl = plen.numpy()[free].astype(np.float32); sm = smooth.numpy()[free].astype(np.float32)
    pl = (pl - pl.min()) / (pl.max() - pl.min()); sm = (sm - sm.min()) / (sm.max() - sm.min())
    tot = np.float32(w_smooth) * sm + np.float32(w_len) * pl
    k = int(np.argmin(tot))
    return int(free.sum()), k, int(np.flatnonzero(free)[k])


def _merge_worker(rank, world, port, counts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    rdist.init_process_group("gloo")
    n_total = sum(counts)
    g = torch.Generator().manual_seed(7)
    traj = torch.randn(n_total, 6, 4, generator=g)
    mask = (torch.rand(n_total, generator=g) < 0.5).to(torch.int32)
    plen = torch.rand(n_total, generator=g) + 1.0
    smooth = torch.rand(n_total, generator=g) + 0.5
    lo = sum(counts[:rank]); hi = lo + counts[rank]
    best, n_free, row = rdist.select_best_sharded(traj[lo:hi], mask[lo:hi], plen[lo:hi], smooth[lo:hi], 0.1, 0.9, select_fn=_costs_select)
    n_ref, _, row_ref = _costs_select(mask, plen, smooth, 0.1, 0.9)            # the unsharded selection
    want = traj[row_ref].clone(); want[0, 2:] = 0.0
    ok = n_free == n_ref and row == row_ref and torch.equal(best, want)
    # nobody is collision-free: every rank learns it, nothing is broadcast
    none, nf, _ = rdist.select_best_sharded(traj[lo:hi], torch.ones_like(mask[lo:hi]), plen[lo:hi], smooth[lo:hi], 0.1, 0.9, select_fn=_costs_select)
    ok = ok and none is None and nf == 0
    # verified gather: the block really holds `world` shards; a tampered block is caught
    if list(counts) == rdist.shard_counts(n_total, world):      # (a rank WITHOUT candidates only exists in the selection: [7, 0])
        local = traj[lo:hi].contiguous()
        gathered = rdist.all_gather_trajectories(local, n_total)
        chk = rdist.verify_gather(local, gathered, n_total)
        bad = gathered.clone(); bad[0, 0, 0] += 1.0
        chk_bad = rdist.verify_gather(local, bad, n_total)
        ok = ok and chk == {"world": world, "ranks_seen": list(range(world)), "checksum_ok": True} and not chk_bad["checksum_ok"]
    # what bench.py --gpus N prints to make a multi-GPU line diagnosable: per-rank job time, the all-gather's own span
    rep = rdist.rank_timing_report(1.0 + rank, 0.25 * (world - rank), 2.0 + 0.5 * rank)
    ok = ok and rep["world"] == world and rep["per_rank_job_s"] == [1.0 + r for r in range(world)] \
        and rep["job_s_min"] == 1.0 and rep["job_s_max"] == float(world) \
        and rep["per_rank_all_gather_s"] == [0.25 * (world - r) for r in range(world)] and rep["all_gather_s_min"] == 0.25 \
        and rep["wall_s_max"] == 2.0 + 0.5 * (world - 1)
    # the lock-step scratch re-plan's decision: the lowest rank that found a plan, or -1 on every rank
    ok = ok and rdist.lowest_rank_with(rank == world - 1, "cpu") == world - 1 and rdist.lowest_rank_with(True, "cpu") == 0 \
        and rdist.lowest_rank_with(False, "cpu") == -1
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("counts", [[5, 5], [4, 3], [7, 0]])
def test_merged_selection_and_verified_gather_world2(counts):
    """Config 4 over several GPUs (diffusion_model_dynamic.py:547, 592-608): candidates sharded, the selection
    merged -- equal to the unsharded compute_trajectory_costs selection (winner row, x[0, 2:] = 0, n_free), also for uneven
    shards and for 'nobody is free'; and the all-gather check bench.py relies on for its n_ranks_rccl."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_gpus_flag_launch_and_refusal_paths():
    """`python bench.py --gpus 2` without torchrun: the invoked process stays GPU-free, starts two ranks with the torchrun
    environment and exits non-zero when they cannot run (here: no GPU -- each rank refuses instead of reporting a smaller
    job); a rank whose WORLD_SIZE differs from --gpus refuses before touching anything."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--scaling", "strong", "--batch", "64"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("2 ranks but only") >= 1 or "needs a HIP device" in r.stderr, r.stderr[-1500:]
    assert "trajectories/s" not in r.stdout                      # no line from a job that did not run
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env2, timeout=300)
    assert r2.returncode != 0 and "refusing to report" in r2.stderr, r2.stderr[-1500:]
