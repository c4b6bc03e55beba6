"""Cost of the skeleton projection loss (train_rig.py:459-470) inside a WHOLE captured training iteration at the bench
workload (300k Gaussians, 24 joints, 800x800): the iteration without the term and with it (learning rates set to 0 so
that both variants time the same frame: the iteration is chaotic otherwise and its cost follows the scene).

    python tools/train_step_projection.py [M]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from riggs_amd.graph import GraphedTrainStep  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402


def measure(thinned):
    sc, cam, gm, sw = bench.build_workload(0, "cuda")
    gm.training_setup(bench._train_args(), capturable=True)
    opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()],
                    lr=0.0, eps=1e-15, capturable=True)
    for o in (gm.optimizer, opt):  # learning rates 0: the scene does not evolve, so both variants time the SAME frame
        for g in o.param_groups:
            if isinstance(g["lr"], torch.Tensor):
                g["lr"].zero_()
            else:
                g["lr"] = 0.0
    torch.manual_seed(0)
    target = torch.rand(3, cam.image_height, cam.image_width, device="cuda")
    gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device="cuda"), target, [gm.optimizer, opt], thinned=thinned)
    gts.capture()
    for _ in range(10):
        gts.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        gts.run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 100 * 1e3
    extra = "" if thinned is None else "  projection loss %.4f" % float(gts.out["projection_loss"])
    return ms, extra


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    torch.manual_seed(1)
    thinned = torch.stack([torch.randint(150, 650, (M,)), torch.randint(150, 650, (M,))], -1).float().cuda()
    for name, th in (("without the projection term", None), ("with the projection term", thinned)):
        ms, extra = measure(th)
        print("%-34s %.4f ms per training iteration (%.0f it/s)%s" % (name, ms, 1e3 / ms, extra), flush=True)


if __name__ == "__main__":
    main()
