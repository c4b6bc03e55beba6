"""The heads-on captured training iteration (bench.train_step_heads_timing's configuration) alone, for a kernel trace:
    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d <out> -- python tools/heads_iteration_profile.py [dense]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from riggs_amd.graph import GraphedFrame, GraphedTrainStep  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402


def main():
    dev, w = "cuda:0", bench.WORKLOAD
    sc, cam, gm, _ = bench.build_workload(0, dev, surface="dense" in sys.argv)
    torch.manual_seed(w["seed"])
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).to(dev).use_fused_heads(True)
    sw._node_radius.data = sc["node_radius"].to(dev)
    gm.training_setup(bench._train_args(), capturable=True)
    opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()], lr=0.0, eps=1e-15,
                    capturable=True)
    bg = torch.zeros(3, device=dev)
    img0 = GraphedFrame(gm, sw, cam, bg, bench.params_of(gm, sw)).capture().run()["render"].detach().clone()
    target = (img0 + 0.05 * torch.randn(img0.shape, generator=torch.Generator().manual_seed(w["seed"] + 7)).to(dev)).clamp_(0.0, 1.0)
    for p in gm.parameters() + list(sw.parameters()):
        p.grad = None
    gts = GraphedTrainStep(gm, sw, cam, bg, target, [gm.optimizer, opt], lambda_dssim=0.2, sparse_grad_rows=True,
                           lambda_template_offsets=1.0, lambda_template_fixed=100.0)
    gts.capture()
    for _ in range(30):
        gts.run()
    torch.cuda.synchronize()
    gts.check()


if __name__ == "__main__":
    main()
