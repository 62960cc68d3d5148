import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from ramp_amd import synth
B = int(sys.argv[1])
bench.WL = bench.WORKLOADS[5]
WL = bench.WL
dm, _ = bench.build_model(B, torch.device("cuda:0"))
dm.use_graph = len(sys.argv) > 2
cloud = torch.from_numpy(synth.make_cloud(WL["cloud"][0], WL["cloud"][1], 3, seed=42)).cuda()
hc = {k: torch.from_numpy(v).cuda() for k, v in synth.default_hard_conds(WL["S"], WL["H"]).items()}
for i in range(3):
    try:
        out = bench.run_job(dm, B, cloud, hc, 1)
        torch.cuda.synchronize()
        print("call", i, "ok", float(out.abs().max()), flush=True)
    except Exception as e:
        print("call", i, "ERR", str(e)[:100], flush=True)
