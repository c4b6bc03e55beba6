"""Graph replays of one configuration of SURVEY.md §8 (C1..C5) for a kernel trace:
  rocprofv3 --kernel-trace -f rocpd -d out -o t -- python tools/config_timeline.py C5 ; python tools/timeline.py out/.../t_results.db"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from configs_sweep import CONFIGS, EXTRA  # noqa: E402
from riggs_amd.graph import GraphedFrame  # noqa: E402

cfg = {**CONFIGS, **EXTRA}[sys.argv[1] if len(sys.argv) > 1 else "C5"]
bench.WORKLOAD.update(cfg)
sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device="cuda"), bench.params_of(gm, sw), sparse_grad_rows=True).capture()  # (as bench.py and configs_sweep.py)
gf.set_inputs(gimg=torch.rand(3, cfg["H"], cfg["W"], device="cuda") * 1e-6)
for _ in range(15):
    gf.run()
torch.cuda.synchronize()
print("R =", gf.check())
