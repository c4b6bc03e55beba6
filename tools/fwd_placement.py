import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from riggs_amd import _lib as L
from riggs_amd.dist import FlatGradAllReduce
from riggs_amd.rasterizer import RasterArena
w = bench.WORKLOAD
sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
T = ((w["W"] + 15) // 16) * ((w["H"] + 15) // 16)
trace = torch.zeros(T * 16 * 6, dtype=torch.int64, device="cuda")
gimg = torch.rand(3, w["H"], w["W"], device="cuda") * 1e-6
step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, FlatGradAllReduce(bench.params_of(gm, sw), register=False))
step()
L.lib().riggs_raster_set_trace(trace.data_ptr())
step(); torch.cuda.synchronize()
L.lib().riggs_raster_set_trace(None)
t = trace.cpu().numpy().reshape(-1, 4, 6)   # [item(tile*4+sub), wave, field]
t = t[t[:, 0, 5] > 0]
hw = (t[:, 0, 2] >> 32) & 0xFFFF
xcc = (t[:, 0, 2] >> 48) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 10 * 0 + cu  # (sh folded)
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
us = t[:, :, 0].max(1) / 100.0
uk, cnt = np.unique(key, return_counts=True)
print("items", len(t), "distinct CUs used", len(uk), "items per CU: min %d max %d mean %.1f" % (cnt.min(), cnt.max(), cnt.mean()))
load = np.zeros(uk.max() + 1); np.add.at(load, key, us)
l = load[uk]
print("sum of item times per CU (us): min %.0f p10 %.0f median %.0f p90 %.0f max %.0f" % (l.min(), *np.percentile(l, [10, 50, 90]), l.max()))
print("per-XCC item-time totals:", [int(us[xcc == x].sum()) for x in range(8)])
idx = np.nonzero(trace.cpu().numpy().reshape(-1, 4, 6)[:, 0, 5] > 0)[0]
for i in np.argsort(-us)[:6]:
    print("item %d (tile %d sub %d): %.1f us rounds %d full %d len %d" % (idx[i], idx[i] // 4, idx[i] % 4, us[i], t[i, 0, 1], t[i, :, 4].max(), t[i, 0, 5]))
top = np.argsort(-us)[:256]
uk2, c2 = np.unique(key[top], return_counts=True)
print("top-256 slowest items sit on %d CUs (max %d per CU)" % (len(uk2), c2.max()))
