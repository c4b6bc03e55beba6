"""Which Python calls put fill / memset nodes into the captured training iteration?  Wraps the usual suspects and prints the
caller while the stream is capturing."""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd.graph import GraphedTrainStep  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            st = [l for l in traceback.format_stack(limit=8) if "find_fills" not in l]
            print("== %s during capture\n%s" % (name, "".join(st[-4:])), flush=True)
        return orig(*a, **k)
    setattr(owner, name, f)


for n in ("zeros", "zeros_like", "full", "ones", "empty_like"):
    if n != "empty_like":
        wrap(torch, n)
for n in ("zero_", "fill_"):
    wrap(torch.Tensor, n)
lib = L.lib()
orig_reset = lib.riggs_raster_binning_reset_history


dev = "cuda:0"
w = bench.WORKLOAD
sc, cam, gm, sw = bench.build_workload(0, dev)
target = torch.rand(3, w["H"], w["W"], generator=torch.Generator().manual_seed(w["seed"] + 100)).to(dev)
gm.training_setup(bench._train_args(), capturable=True)
sk_opt = FusedAdam([{"params": g_["params"], "lr": 5e-4, "name": g_["name"]} for g_ in sw.trainable_parameters()], lr=0.0, eps=1e-15,
                   capturable=True)
gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device=dev), target, [gm.optimizer, sk_opt], lambda_dssim=0.2, sparse_grad_rows=True)
gts.capture()
print("captured")
