"""Node list of the captured training iteration (hipGraph debug dump): which nodes are memsets, and between which kernels."""
import os
import re
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from riggs_amd.graph import GraphedTrainStep  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402

orig = torch.cuda.CUDAGraph


class G(orig):
    def __new__(cls, *a, **k):
        g = super().__new__(cls)
        g.enable_debug_mode()
        return g


torch.cuda.CUDAGraph = G
dev = "cuda:0"
w = bench.WORKLOAD
sc, cam, gm, sw = bench.build_workload(0, dev)
target = torch.rand(3, w["H"], w["W"], generator=torch.Generator().manual_seed(w["seed"] + 100)).to(dev)
gm.training_setup(bench._train_args(), capturable=True)
sk_opt = FusedAdam([{"params": g_["params"], "lr": 5e-4, "name": g_["name"]} for g_ in sw.trainable_parameters()], lr=0.0, eps=1e-15,
                   capturable=True)
gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device=dev), target, [gm.optimizer, sk_opt], lambda_dssim=0.2, sparse_grad_rows=True)
gts.capture()
path = "/tmp/train_graph.dot"
gts.graph.debug_dump(path)
txt = open(path).read()
print(len(txt), "bytes")
for line in txt.splitlines():
    m = re.search(r'label="([^"]*)"', line)
    if m and "->" not in line:
        print(m.group(1)[:200].replace("\\n", " | "))
