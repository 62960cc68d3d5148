"""Eager score evaluations (forward + dX backward, CFG rows) of the headline workload, for rocprofv3 --pmc passes.  In the
default fp16x3 mode the first evaluation calibrates on the bf16x6 kernels; the summaries use the LAST pass (fp16x3, NP = 2).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_score_fetch -- python3 ramp_amd/tools/score_pmc.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_score_write -- python3 ramp_amd/tools/score_pmc.py
    python3 ramp_amd/tools/pmc_summary.py gpurun_out/pmc_score_fetch gpurun_out/pmc_score_write profiles/rNN_pmc_traffic.json
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
from ramp_amd import synth  # noqa: E402

cloud = torch.from_numpy(synth.make_cloud(*bench.WL["cloud"], 2, seed=42)).cuda()
x = torch.randn(B, bench.WL["H"], bench.WL["S"], device="cuda")
t = torch.full((B,), 12, dtype=torch.long, device="cuda")
dm.ddim = False
for _ in range(3):          # pass 1 calibrates (bf16x6), passes 2-3 run the timed fp16x3 kernels; the summary uses the last
    print("score pass", flush=True)
    dm.p_mean_variance(x, None, None, t, obstacle_pts=cloud)
    torch.cuda.synchronize()
    print("  arithmetic:", dm.model.score_mode(), flush=True)
assert dm.model.score_mode() == "fp16x3"
print("done", flush=True)
