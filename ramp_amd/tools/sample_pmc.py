"""One eager 3-step sampling job (ramp_sample, B = 4096 x 2 CFG rows, shared prefix, fused feed-forward) of the headline
workload, for rocprofv3 --pmc passes: evaluation 0 calibrates on the bf16x6 kernels, evaluations 1-2 are the fp16x3
evaluations the bench times; the summaries use the LAST one (the dispatches between the last two cfg_mean launches).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_sample_fetch -- python3 ramp_amd/tools/sample_pmc.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_sample_write -- python3 ramp_amd/tools/sample_pmc.py
    python3 ramp_amd/tools/pmc_summary.py gpurun_out/pmc_sample_fetch gpurun_out/pmc_sample_write profiles/rNN_pmc_traffic.json
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bench.WL = bench.WORKLOADS[2]
dm, _ = bench.build_model(B, torch.device("cuda:0"))
dm.use_graph = False
from ramp_amd import synth  # noqa: E402

cloud = torch.from_numpy(synth.make_cloud(*bench.WL["cloud"], 2, seed=42)).cuda()
H, S = bench.WL["H"], bench.WL["S"]
hard_conds = {k: torch.from_numpy(v).cuda().unsqueeze(0).expand(B, -1).contiguous() for k, v in synth.default_hard_conds(S, H).items()}
noise = torch.randn(4, B, H, S, device="cuda")
out, _ = dm._launch(B, noise, hard_conds, cloud, False, [24, 23, 22], [0, 0, 0], [0.5, 0.5, 0.5], None, False)   # the job's first three steps
torch.cuda.synchronize()
assert bool(torch.isfinite(out).all())
print("done", flush=True)
