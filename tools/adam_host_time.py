"""Host time of the eagerly issued optimizer step of a training iteration (Gaussians' + skeleton's FusedAdam, train_rig.py:527-554)
with the cached launch plan and with the plan rebuilt every step (what every step cost before).  usage: python tools/adam_host_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd.optim import FusedAdam, step_many  # noqa: E402

sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
gm.training_setup(bench._train_args())
sk = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()], lr=0.0, eps=1e-15)
params = [p for o in (gm.optimizer, sk) for g in o.param_groups for p in g["params"]]


def run(rebuild, n=300):
    times = []
    for it in range(n + 20):
        for p in params:
            p.grad = torch.zeros_like(p)          # (new gradient objects every iteration, as zero_grad(set_to_none=True) leaves them)
        if rebuild:
            gm.optimizer._hip_plan = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step_many([gm.optimizer, sk])
        t1 = time.perf_counter()
        if it >= 20:
            times.append(t1 - t0)
    times.sort()
    return 1e6 * times[len(times) // 2]


print("tensors: %d" % len(params))
print("host time of the merged step, median: cached plan %.1f us | rebuilt every step %.1f us" % (run(False), run(True)))
