"""Host profile of the eagerly issued TRAINING iteration (train_rig.py:411-554 as an unmodified trainer runs it)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from riggs_amd.loss import l1_loss, ssim  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402
from riggs_amd.render import render  # noqa: E402

dev = "cuda:0"
w = bench.WORKLOAD
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
acc = {}
pc = time.perf_counter


def it():
    t = pc()
    gm.optimizer.zero_grad(set_to_none=True)
    sk.zero_grad(set_to_none=True)
    t1 = pc(); acc["zero_grad"] = acc.get("zero_grad", 0) + t1 - t
    dv = sw(gm.get_xyz.detach(), t_in, motion_mask=gm.motion_mask)
    pkg = render(cam, gm, bench.Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"], arena=arena)
    t2 = pc(); acc["deform+render"] = acc.get("deform+render", 0) + t2 - t1
    loss = 0.8 * l1_loss(pkg["render"], target) + 0.2 * (1.0 - ssim(pkg["render"], target))
    t3 = pc(); acc["loss"] = acc.get("loss", 0) + t3 - t2
    loss.backward()
    t4 = pc(); acc["backward"] = acc.get("backward", 0) + t4 - t3
    gm.optimizer.step()
    sk.step()
    t5 = pc(); acc["optimizers"] = acc.get("optimizers", 0) + t5 - t4


for _ in range(30):
    it()
torch.cuda.synchronize()
acc.clear()
n = 200
t0 = pc()
for _ in range(n):
    it()
t1 = pc()
torch.cuda.synchronize()
print("host loop %.1f us / iteration (with sync %.1f)" % ((t1 - t0) / n * 1e6, (pc() - t0) / n * 1e6))
for k, v in acc.items():
    print("  %-16s %.1f us" % (k, v / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    gm.optimizer.zero_grad(set_to_none=True)
    sk.zero_grad(set_to_none=True)
    dv = sw(gm.get_xyz.detach(), t_in, motion_mask=gm.motion_mask)
    pkg = render(cam, gm, bench.Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"], arena=arena)
    loss = 0.8 * l1_loss(pkg["render"], target) + 0.2 * (1.0 - ssim(pkg["render"], target))
    loss.backward()
    gm.optimizer.step()
    sk.step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

# ---- the backward by node: own time of the three HIP-backed backward functions; the rest is the engine, the scalar glue
# of "0.8 * l1 + 0.2 * (1 - ssim)" and 33 AccumulateGrad nodes
from riggs_amd import loss as LO, render as RE, skeleton as SK  # noqa: E402
bacc = {}


def wrapb(cls, key):
    orig = cls.backward

    def f(*a, **k):
        t = pc()
        r = orig(*a, **k)
        bacc[key] = bacc.get(key, 0.0) + pc() - t
        return r
    cls.backward = staticmethod(f)


wrapb(LO._L1SSIM, "loss.backward")
wrapb(RE._FusedGlueRaster, "raster.backward")
wrapb(SK._PoseDeform, "posedeform.backward")
acc.clear()
for _ in range(n):
    it()
torch.cuda.synchronize()
print("backward wall %.1f us; inside the three functions:" % (acc["backward"] / n * 1e6))
for k, v in bacc.items():
    print("  %-22s %.1f us" % (k, v / n * 1e6))
