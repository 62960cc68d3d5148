"""Multi-GPU: one process per GPU, the trajectory-sample batch sharded contiguously across ranks.

Every trajectory is independent through the whole reverse-diffusion loop (per-sample norms, per-sample
energy gradient; weights, schedule and scene latent are replicated read-only), so the data path has NO
collective.  The single exchange is one all-gather of the final (B/G, H, S) trajectories at the end
(RCCL over xGMI on the GPU box: backend "nccl"; "gloo" in the CPU tests) — SURVEY.md §8(e).
The reference has no inference-time parallelism at all (its only distributed code is DDP training,
scripts/train/trainddp.py), so there is no reference call pattern to mirror here.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend: str) -> Tuple[int, int, int]:
    rank, world, local = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_counts(n_total: int, world: int) -> List[int]:
    """Contiguous shards, sizes differ by at most one (first ``n_total % world`` ranks get the extra)."""
    base, extra = divmod(n_total, world)
    return [base + (1 if r < extra else 0) for r in range(world)]


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    counts = shard_counts(n_total, world)
    start = sum(counts[:rank])
    return start, start + counts[rank]


def all_gather_trajectories(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """local (b_rank, H, S) -> (n_total, H, S) on every rank, ordered by global sample index."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert local.shape[0] == n_total
        return local
    world = dist.get_world_size()
    counts = shard_counts(n_total, world)
    assert local.shape[0] == counts[dist.get_rank()], "local shard does not match shard_counts()"
    if len(set(counts)) == 1:
        out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    pad = max(counts)
    buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def sample_sharded(sample_local: Callable[[int, int], torch.Tensor], n_total: int, gather: bool = True) -> torch.Tensor:
    """Run ``sample_local(start, stop)`` (returns the final trajectories of global samples [start, stop))
    on this rank's shard and all-gather the results."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    start, stop = shard_range(n_total, rank, world)
    local = sample_local(start, stop)
    return all_gather_trajectories(local, n_total) if gather else local


def verify_gather(local: torch.Tensor, gathered: torch.Tensor, n_total: int) -> Dict[str, object]:
    """Proof that a gathered block really came from ``world`` ranks: every rank contributes (rank, float64 checksum of its
    local shard) through a second, real all-gather; the block must contain exactly those shards in rank order.  Returns
    {"world": size read back from the initialised process group, "ranks_seen": [...], "checksum_ok": bool}.
    A silent single-rank run (WORLD_SIZE lost, a launcher that started one process) cannot produce world > 1 here."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        ok = bool(torch.equal(local, gathered))
        return {"world": 1, "ranks_seen": [0], "checksum_ok": ok}
    world, rank = dist.get_world_size(), dist.get_rank()
    tag = torch.tensor([float(rank), float(local.double().sum().item())], dtype=torch.float64, device=local.device)
    tags = torch.empty(world * 2, dtype=torch.float64, device=local.device)      # (concatenation along dim 0: the form every backend takes)
    dist.all_gather_into_tensor(tags, tag)
    tags = tags.reshape(world, 2).cpu()
    counts = shard_counts(n_total, world)
    ok, start = True, 0
    for r in range(world):
        part = gathered[start:start + counts[r]].double().sum().item()
        ref = float(tags[r, 1])
        ok = ok and abs(part - ref) <= 1e-9 * max(1.0, abs(ref))
        start += counts[r]
    return {"world": world, "ranks_seen": [int(v) for v in tags[:, 0].tolist()], "checksum_ok": bool(ok)}


def select_best_sharded(batch_local: torch.Tensor, mask_local: torch.Tensor, plen_local: torch.Tensor,
                        smooth_local: torch.Tensor, w_smooth: float, w_len: float,
                        select_fn: Optional[Callable] = None, zero_start: bool = True):
    """The planner's selection (compute_trajectory_costs, cost.py:56-88: min-max normalised 0.1 smoothness + 0.9 length
    over the collision-free candidates, first minimum) when the candidates are sharded over the ranks: all-gather the three
    per-candidate scalars (12 bytes per candidate), run the SAME selection on the gathered arrays on every rank, and let the
    rank that owns the winner broadcast its (H, S) trajectory.  Returns (best, n_free, global_row); best is None when no
    candidate is collision-free.  select_fn(mask, plen, smooth, w_smooth, w_len) -> (n_free, best_rank, best_row); the
    default is the HIP selection kernel (ramp_select_from_costs), the CPU tests pass the oracle's restatement."""
    if select_fn is None:
        select_fn = _select_hip
    sharded = dist.is_initialized() and dist.get_world_size() > 1
    world, rank = (dist.get_world_size(), dist.get_rank()) if sharded else (1, 0)
    n_local = int(mask_local.shape[0])
    if sharded:
        cnt = torch.tensor([n_local], dtype=torch.int64, device=mask_local.device)
        cnts = torch.empty(world, dtype=torch.int64, device=mask_local.device)
        dist.all_gather_into_tensor(cnts, cnt)
        counts = [int(v) for v in cnts.cpu()]
        pad = max(counts)
        pack = torch.zeros((pad, 3), dtype=torch.float32, device=mask_local.device)
        pack[:n_local, 0] = mask_local.to(torch.float32); pack[:n_local, 1] = plen_local; pack[:n_local, 2] = smooth_local
        allp = torch.empty((world * pad, 3), dtype=torch.float32, device=mask_local.device)
        dist.all_gather_into_tensor(allp, pack)
        allp = allp.reshape(world, pad, 3)
        parts = torch.cat([allp[r, :counts[r]] for r in range(world)], dim=0)
        mask_all = parts[:, 0].to(torch.int32).contiguous(); plen_all = parts[:, 1].contiguous(); smooth_all = parts[:, 2].contiguous()
    else:
        counts = [n_local]
        mask_all, plen_all, smooth_all = mask_local.to(torch.int32).contiguous(), plen_local.contiguous(), smooth_local.contiguous()
    n_free, _best_rank, row = select_fn(mask_all, plen_all, smooth_all, w_smooth, w_len)
    if n_free == 0:
        return None, 0, -1
    owner, start = 0, 0
    while row >= start + counts[owner]:
        start += counts[owner]; owner += 1
    # (the receive buffer is shaped from (H, S) and the dtype, not from batch_local[0]: a rank may hold no candidate at all)
    best = batch_local[row - start].clone() if owner == rank else torch.empty(tuple(batch_local.shape[1:]), dtype=batch_local.dtype,
                                                                                  device=batch_local.device)
    if sharded:
        dist.broadcast(best, src=owner)
    if zero_start:
        best[0, 2:] = 0.0                                     # diffusion_model_dynamic.py:607
    return best, int(n_free), int(row)


def rank_timing_report(job_s: float, gather_s: float, wall_s: float, device=None) -> dict:
    """What makes a multi-GPU bench line diagnosable: every rank's own sampling-job time, its all-gather span (which contains
    the wait for the slowest rank) and its wall time of the timed region, gathered with one small collective AFTER the timed
    region.  Without a process group the lists have one entry."""
    vals = torch.tensor([job_s, gather_s, wall_s], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        allv = torch.empty(dist.get_world_size() * 3, dtype=torch.float64, device=vals.device)
        dist.all_gather_into_tensor(allv, vals)
        allv = allv.reshape(-1, 3)
    else:
        allv = vals[None]
    allv = allv.cpu()
    job, gat, wall = ([float(v) for v in allv[:, k]] for k in range(3))
    return {"world": int(allv.shape[0]),
            "per_rank_job_s": job, "job_s_min": min(job), "job_s_max": max(job),
            "per_rank_all_gather_s": gat, "all_gather_s_min": min(gat), "all_gather_s_max": max(gat),
            "per_rank_wall_s": wall, "wall_s_max": max(wall),
            "note": "job = HIP events around run_inference on the rank's stream; all-gather = from the rank's own job end to the "
                    "collective's end (its minimum over ranks is the collective itself, the rest is waiting for slower ranks)"}


def lowest_rank_with(flag: bool, device) -> int:
    """The lowest rank whose `flag` is set, or -1 when no rank's is (one MIN all-reduce of a rank number); without a process
    group: 0 / -1.  Used by the sharded planner's lock-step scratch re-plan."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return 0 if flag else -1
    world = dist.get_world_size()
    t = torch.tensor([dist.get_rank() if flag else world], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    v = int(t.item())
    return v if v < world else -1


def min_over_ranks(value: int, device) -> int:
    """The minimum of `value` over the ranks (one MIN all-reduce); `value` itself without a process group.  The sharded
    planner sizes its scratch re-plan with it, so that every rank draws the same number of random values per round."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def _select_hip(mask, plen, smooth, w_smooth, w_len):
    from . import _lib
    res = torch.zeros(4, dtype=torch.int32, device=mask.device)
    with torch.cuda.device(mask.device):
        _lib.check(_lib.load().ramp_select_from_costs(_lib.ptr(mask), _lib.ptr(plen), _lib.ptr(smooth), int(mask.shape[0]),
                                                      float(w_smooth), float(w_len), _lib.ptr(res), _lib.current_stream()),
                   "ramp_select_from_costs")
    n_free, best_rank, row, _ = (int(v) for v in res.cpu())
    return n_free, best_rank, row


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_local_ranks(argv: List[str], world: int, extra_env: Optional[Dict[str, str]] = None,
                       timeout: Optional[float] = 3600.0) -> int:
    """Start ``world`` copies of ``python argv...`` on this node, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set the way torchrun sets them, and wait for all of them.  The caller is a parent that has
    NOT initialised the GPU (it only spawns children and relays rank 0's stdout).  Returns 0 only if every rank exited
    with 0; when one rank fails the others are terminated (a rank stuck in a collective would otherwise hang forever)."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    import time
    t0 = time.time()
    live = list(procs)
    try:
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:
                        q.terminate()
            if timeout is not None and time.time() - t0 > timeout and live:
                rc = 124                                       # a rank hung (e.g. in a rendezvous with too few GPUs)
                break
            time.sleep(0.05)
    finally:                                                   # also on KeyboardInterrupt / SIGTERM of the parent: no orphan ranks
        for q in procs:
            if q.poll() is None:
                q.terminate()
        for q in procs:
            try:
                q.wait(timeout=10)
            except Exception:                                  # noqa: BLE001
                q.kill()
    return rc
