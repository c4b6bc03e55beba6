"""A whole training iteration with both MLP heads on (the shipped stage-2 recipe), as one hipGraph, fp32 heads vs the fused
bf16-MFMA heads: deform (PoseMLP, FK, heads, LBS) -> render -> L1+SSIM -> backward -> FusedAdam (Gaussians, skeleton incl. heads).
usage: python tools/heads_train_step.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.gaussian_model import GaussianModel  # noqa: E402
from riggs_amd.graph import GraphedTrainStep  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402


def main():
    w = bench.WORKLOAD
    dev = "cuda:0"
    for fused in (False, True):
        sc = synth.make_scene(w["N"], w["J"], w["seed"])
        cam = synth.look_at_camera(w["H"], w["W"], fid=0.37).to(dev)
        gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                        sc["opacity"], device=dev)
        torch.manual_seed(w["seed"])
        sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).to(dev)
        sw._node_radius.data = sc["node_radius"].to(dev)
        sw.use_fused_heads(fused)
        target = torch.rand(3, w["H"], w["W"], device=dev)
        gm.training_setup(bench._train_args(), capturable=True)
        opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()],
                        lr=0.0, eps=1e-15, capturable=True)
        wref = sw.skinning_weight_mlp.linear[3].weight.detach().clone()
        gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device=dev), target, [gm.optimizer, opt], lambda_dssim=0.2)
        gts.capture()
        early = []
        for _ in range(3):
            gts.run()
            early.append(float(gts.out["loss"]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            gts.run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        moved = float((sw.skinning_weight_mlp.linear[3].weight.detach() - wref).abs().max())
        # (the losses of the first replays are comparable between the two paths; later the iteration — Adam at the reference's
        # learning rates against a random target image — is chaotic: float-atomics noise decides which way it goes)
        print("heads %s: %.2f ms per training iteration (%.1f it/s), loss of replays 1-3 %s, WeightMLP layer-3 weights moved by %.2e"
              % ("fused MFMA (fp16 operands)" if fused else "fp32 GEMMs", dt * 1e3, 1.0 / dt, " ".join("%.4f" % v for v in early), moved))


if __name__ == "__main__":
    main()
