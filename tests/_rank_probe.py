"""Child of tests/test_dist_gloo.py::test_launch_local_ranks: what bench.py's ranks do around the timed region
(rendezvous from the torchrun-style environment, barrier, MAX-over-ranks reduction, one line from rank 0)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from ramp_amd import dist as rdist

rank, world, local = rdist.init_process_group("gloo")
if len(sys.argv) > 1 and sys.argv[1] == "fail" and rank == 1:
    sys.exit(3)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out = rdist.all_gather_trajectories(torch.full((2, 4, 4), float(rank)), 2 * world)
dist.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": world, "max": float(t), "gathered": [float(v) for v in out[:, 0, 0]]}), flush=True)
dist.destroy_process_group()
