"""Per-chunk timeline of the compositing backward on the bench workload.
usage: python tools/bwd_trace.py [dense]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd.dist import FlatGradAllReduce  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402

w = bench.WORKLOAD
sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "dense":
    from riggs_amd import synth
    from riggs_amd.gaussian_model import GaussianModel
    sc = synth.make_surface_scene(w["N"], w["J"], w["seed"])
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"], sc["opacity"], device="cuda:0")
T = ((w["W"] + 15) // 16) * ((w["H"] + 15) // 16)
NB = 1 << 16  # room for the backward's chunks
trace = torch.zeros(T * 256 + NB * 4, dtype=torch.int64, device="cuda")
gimg = torch.rand(3, w["H"], w["W"], device="cuda") * 1e-6
step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, FlatGradAllReduce(bench.params_of(gm, sw), register=False))
step()
L.lib().riggs_raster_set_trace(trace.data_ptr())
step()
torch.cuda.synchronize()
L.lib().riggs_raster_set_trace(None)
b = trace.cpu().numpy()[T * 256:].reshape(-1, 4)
b = b[b[:, 1] > 0]
t0 = b[:, 0].min()
start = (b[:, 0] - t0) / 100.0
end = (b[:, 1] - t0) / 100.0
dur = end - start
wg = b[:, 3] >> 32
chunk = b[:, 3] & 0xFFFF
hw = b[:, 2] & 0xFFFF
xcc = (b[:, 2] >> 16) & 0xF
cu = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xF)
print("chunks %d, workgroups %d, CUs %d; kernel span %.1f us" % (len(b), len(np.unique(wg)), len(np.unique(cu)), end.max()))
print("chunk duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
print("sum of durations / (span * CUs): %.2f chunks in flight per CU on average" % (dur.sum() / (end.max() * len(np.unique(cu)))))
for lo, hi in [(0, 1), (1, 4), (4, 16), (16, 64), (64, 1 << 16)]:
    m = (chunk >= lo) & (chunk < hi)
    if m.any():
        print("  chunk index [%d, %d): n %5d mean %.1f us" % (lo, hi, m.sum(), dur[m].mean()))
first = np.array([start[wg == g].min() for g in np.unique(wg)])
last = np.array([end[wg == g].max() for g in np.unique(wg)])
print("workgroup first start: p50 %.1f p90 %.1f max %.1f; last end: p10 %.1f p50 %.1f max %.1f" % (*np.percentile(first, [50, 90]), first.max(), *np.percentile(last, [10, 50]), last.max()))
h, _ = np.histogram(end, bins=np.arange(0, end.max() + 10, 10))
print("chunks finishing per 10 us:", h.tolist())
