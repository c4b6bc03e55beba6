"""Time the deformation with both per-Gaussian MLP heads on (functional half: the MLPs run as torch
Linear layers = hipBLASLt GEMMs, the skinning kernels take their outputs) at the bench size.
usage: python tools/heads_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402


def main():
    w = bench.WORKLOAD
    sc, cam, gm, sw0 = bench.build_workload(0, "cuda:0")
    J = w["J"]
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).cuda()
    x = gm.get_xyz.detach()
    q = torch.nn.functional.normalize(torch.randn(J, 4, device="cuda"), dim=-1).requires_grad_(True)
    gt = torch.zeros(3, device="cuda", requires_grad=True)
    gx, gr = torch.randn_like(x), torch.randn(x.shape[0], 4, device="cuda")
    params = [p for g in sw.trainable_parameters() for p in g["params"]]

    def it():
        for p in params + [q, gt]:
            p.grad = None
        out = sw.deform_by_pose(x, {"local_rotation": q, "global_trans": gt}, None)
        torch.autograd.backward((out["d_xyz"], out["d_rotation"]), (gx, gr))
    res = {}
    for fused in (False, True):
        sw.use_fused_heads(fused)
        for _ in range(3):
            it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            it()
        torch.cuda.synchronize()
        res[fused] = (time.perf_counter() - t0) / n
    dt = res[False]
    mlp_flops = 0
    for name in ("skinning_weight_mlp", "detail_net"):
        m = getattr(sw, name)
        mlp_flops += sum(2 * p.numel() for n_, p in m.named_parameters() if p.dim() == 2)
    print("deform_by_pose fwd+bwd with WeightMLP + DeformMLP heads, N=%d: %.2f ms per iteration (%.1f TFLOP/s on the MLPs' "
          "3 x %.2f MFLOP per Gaussian)" % (x.shape[0], dt * 1e3, 3 * mlp_flops * x.shape[0] / dt / 1e12, mlp_flops / 1e6))
    dt = res[True]
    print("same with the fused MFMA heads, fp16 operands (riggs_amd.mlp): %.2f ms per iteration (%.1f TFLOP/s)" % (dt * 1e3, 3 * mlp_flops * x.shape[0] / dt / 1e12))


if __name__ == "__main__":
    main()
