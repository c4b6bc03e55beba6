"""torch.profiler (CPU side) over eagerly issued training iterations: which ops / autograd nodes the host time goes to."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from riggs_amd.loss import l1_loss, ssim  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402
from riggs_amd.render import render  # noqa: E402

dev = "cuda:0"
sc, cam, gm, sw = bench.build_workload(0, dev)
bg = torch.zeros(3, device=dev)
t_in = sw.expand_time(cam.fid)
arena = RasterArena()
with torch.no_grad():
    dv0 = sw(gm.get_xyz.detach(), t_in, motion_mask=gm.motion_mask)
    img0 = render(cam, gm, bench.Pipe, bg, dv0["d_xyz"], dv0["d_rotation"], dv0["d_scaling"])["render"]
target = (img0 + 0.05 * torch.randn(img0.shape, generator=torch.Generator().manual_seed(7)).to(dev)).clamp_(0.0, 1.0)
gm.training_setup(bench._train_args())
sk = FusedAdam([{"params": g_["params"], "lr": 5e-4, "name": g_["name"]} for g_ in sw.trainable_parameters()], lr=0.0, eps=1e-15)


def it():
    gm.optimizer.zero_grad(set_to_none=True)
    sk.zero_grad(set_to_none=True)
    dv = sw(gm.get_xyz.detach(), t_in, motion_mask=gm.motion_mask)
    pkg = render(cam, gm, bench.Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"], arena=arena)
    loss = 0.8 * l1_loss(pkg["render"], target) + 0.2 * (1.0 - ssim(pkg["render"], target))
    loss.backward()
    gm.optimizer.step()
    sk.step()


for _ in range(30):
    it()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(50):
        it()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=60))
