"""Where do the forward compositing kernel's workgroups run, and which pixel blocks are its critical path?
usage: python tools/fwd_placement.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd.dist import FlatGradAllReduce  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402

BPT = 8  # pixel blocks (workgroups) per tile; 4 waves each
w = bench.WORKLOAD
sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
T = ((w["W"] + 15) // 16) * ((w["H"] + 15) // 16)
trace = torch.zeros(T * BPT * 4 * 6, dtype=torch.int64, device="cuda")
gimg = torch.rand(3, w["H"], w["W"], device="cuda") * 1e-6
step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, FlatGradAllReduce(bench.params_of(gm, sw), register=False))
step()
L.lib().riggs_raster_set_trace(trace.data_ptr())
step()
torch.cuda.synchronize()
L.lib().riggs_raster_set_trace(None)
full = trace.cpu().numpy().reshape(-1, 4, 6)  # [block (tile * BPT + sub), wave, field]
idx = np.nonzero(full[:, 0, 5] > 0)[0]
t = full[idx]
hw = (t[:, 0, 2] >> 32) & 0xFFFF
xcc = (t[:, 0, 2] >> 48) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
us = t[:, :, 0].max(1) / 100.0
uk, cnt = np.unique(key, return_counts=True)
print("blocks", len(t), "distinct CUs used", len(uk), "blocks per CU: min %d max %d mean %.1f" % (cnt.min(), cnt.max(), cnt.mean()))
load = np.zeros(uk.max() + 1)
np.add.at(load, key, us)
l = load[uk]
print("sum of block times per CU (us): min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (l.min(), *np.percentile(l, [10, 50, 90]), l.max()))
for i in np.argsort(-us)[:6]:
    print("block %d (tile %d sub %d): %.1f us rounds %d steps %s with-contribution %s survivors %s list %d" % (
        idx[i], idx[i] // BPT, idx[i] % BPT, us[i], t[i, 0, 1], t[i, :, 3].tolist(), t[i, :, 4].tolist(),
        (t[i, :, 2] & 0xFFFFFFFF).tolist(), t[i, 0, 5]))
