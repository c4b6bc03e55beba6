"""Kernel trace helper: a few iterations of the deformation with the FUSED heads only (tools/heads_bench.py runs both variants).
  rocprofv3 --kernel-trace --stats -f csv -d out -o t -- python tools/scratch/heads_trace.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402

w = bench.WORKLOAD
sc, cam, gm, sw0 = bench.build_workload(0, "cuda:0")
J = w["J"]
sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).cuda()
x = gm.get_xyz.detach()
q = torch.nn.functional.normalize(torch.randn(J, 4, device="cuda"), dim=-1).requires_grad_(True)
gt = torch.zeros(3, device="cuda", requires_grad=True)
gx, gr = torch.randn_like(x), torch.randn(x.shape[0], 4, device="cuda")
params = [p for g in sw.trainable_parameters() for p in g["params"]]
sw.use_fused_heads(True)


def it():
    for p in params + [q, gt]:
        p.grad = None
    out = sw.deform_by_pose(x, {"local_rotation": q, "global_trans": gt}, None)
    torch.autograd.backward((out["d_xyz"], out["d_rotation"]), (gx, gr))


for _ in range(3):
    it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    it()
torch.cuda.synchronize()
print("fused heads: %.3f ms per iteration" % ((time.perf_counter() - t0) / 10 * 1e3))
