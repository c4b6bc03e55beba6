"""Cost of the ordered-reduction mode of the compositing backward (riggs_raster_cfg.deterministic) against the float-atomics
path, at the bench workload: eager rasterizer forward + backward, events around the backward.
usage: python tools/ordered_bwd_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import rasterizer as RZ, synth  # noqa: E402
from tests import gpu_util as U  # noqa: E402

w = bench.WORKLOAD
sc, act, cam = U.activated_scene(w["N"], w["J"], w["seed"], w["H"], w["W"])
d = lambda t: t.cuda().contiguous()  # noqa: E731
args = (d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]), None)
gc = (torch.sign(torch.rand(3, w["H"], w["W"]) - 0.5) / (3 * w["H"] * w["W"])).cuda()
for mode in (False, True):
    RZ.set_ordered_backward(mode)
    ts = []
    for it in range(12):
        out = RZ.rasterize_forward(U.settings_for(cam, [0, 0, 0]), *args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        RZ.rasterize_backward(out[4], *args, None, None, gc, None, None)
        b.record()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append(a.elapsed_time(b))
    ts.sort()
    print("%s: rasterizer backward (compositing + per-Gaussian) median %.3f ms" % ("ordered rows + gather" if mode else "float atomics", ts[len(ts) // 2]))
RZ.set_ordered_backward(False)
