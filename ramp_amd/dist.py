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


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_local_ranks(argv: List[str], world: int, extra_env: Optional[Dict[str, str]] = None,
                       timeout: Optional[float] = None) -> int:
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
            for q in live:
                q.kill()
            return 124
        time.sleep(0.05)
    return rc
