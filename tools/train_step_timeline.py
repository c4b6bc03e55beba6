"""Graph replays of one WHOLE training iteration (bench.py's `train_step`: the metric's path + fused image loss + FusedAdam steps)
for a kernel trace:
  rocprofv3 --kernel-trace -f rocpd -d out -o t -- python tools/train_step_timeline.py ; python tools/timeline.py out/.../t_results.db"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd.graph import GraphedTrainStep  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402

dev = "cuda:0"
w = bench.WORKLOAD
sc, cam, gm, sw = bench.build_workload(0, dev)
# (target = the scene's own render plus noise, as bench.py's train_step: against a uniform-random image Adam walks every scale up
# and the instance count triples within a hundred iterations)
from riggs_amd.render import render  # noqa: E402
with torch.no_grad():
    dv0 = sw(gm.get_xyz.detach(), sw.expand_time(cam.fid), motion_mask=gm.motion_mask)
    img0 = render(cam, gm, bench.Pipe, torch.zeros(3, device=dev), dv0["d_xyz"], dv0["d_rotation"], dv0["d_scaling"])["render"]
target = (img0 + 0.05 * torch.randn(img0.shape, generator=torch.Generator().manual_seed(w["seed"] + 7)).to(dev)).clamp_(0.0, 1.0)
gm.training_setup(bench._train_args(), capturable=True)
sk_opt = FusedAdam([{"params": g_["params"], "lr": 5e-4, "name": g_["name"]} for g_ in sw.trainable_parameters()], lr=0.0, eps=1e-15,
                   capturable=True)
gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device=dev), target, [gm.optimizer, sk_opt], lambda_dssim=0.2, sparse_grad_rows=True)
gts.capture()
for _ in range(5):
    gts.run()
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(30):
    gts.run()
torch.cuda.synchronize()
print("train step: %.4f ms, R = %d" % ((time.perf_counter() - t1) / 30 * 1e3, gts.check()))
