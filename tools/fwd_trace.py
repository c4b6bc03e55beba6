"""Per-wave statistics of the forward compositing kernel on the bench workload: where does its time go?
usage: python tools/fwd_trace.py [dense]     (dense: the dense-gradient scene of riggs_amd.synth.make_surface_scene)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd.dist import FlatGradAllReduce  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402


def main():
    w = bench.WORKLOAD
    sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    if len(sys.argv) > 1 and sys.argv[1] == "dense":
        from riggs_amd import synth
        from riggs_amd.gaussian_model import GaussianModel
        sc = synth.make_surface_scene(w["N"], w["J"], w["seed"])
        gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                        sc["opacity"], device="cuda:0")
    T = ((w["W"] + 15) // 16) * ((w["H"] + 15) // 16)
    BPT = 8  # pixel blocks (workgroups) per tile; 4 waves each
    trace = torch.zeros(T * BPT * 4 * 6, dtype=torch.int64, device="cuda")
    gimg = torch.rand(3, w["H"], w["W"], device="cuda") * 1e-6
    step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, FlatGradAllReduce(bench.params_of(gm, sw), register=False))
    step()
    L.lib().riggs_raster_set_trace(trace.data_ptr())
    step()
    torch.cuda.synchronize()
    L.lib().riggs_raster_set_trace(None)
    t = trace.cpu().numpy().reshape(-1, 6)
    t = t[t[:, 5] > 0]
    us = t[:, 0] / 100.0
    order = np.argsort(-us)
    print("waves with work: %d; time us: max %.1f p99 %.1f p90 %.1f median %.1f" % (len(t), us.max(), *np.percentile(us, [99, 90, 50])))
    print("slowest waves: [us, rounds, survivors/4waves, iterations, full iterations, list length]")
    for i in order[:12]:
        print("  %7.1f %5d %7d %6d %6d %7d" % (us[i], t[i, 1], t[i, 2] & 0xFFFFFFFF, t[i, 3], t[i, 4], t[i, 5]))
    # simple linear model of a wave's time
    A = np.stack([np.ones(len(t)), t[:, 1], t[:, 3] - t[:, 4], t[:, 4]], 1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, us, rcond=None)
    print("fit: us = %.2f + %.3f*rounds + %.3f*skipped_iterations + %.3f*full_iterations" % tuple(coef))
    per_tile = t[::1]
    lens = np.unique(np.stack([np.arange(len(trace) // 6)[trace.cpu().numpy().reshape(-1, 6)[:, 5] > 0] // (BPT * 4), t[:, 5]], 1), axis=0)[:, 1]
    print("non-empty tiles: %d; list length: median %d, p90 %d, max %d; lists <= 256: %d, <= 512: %d, <= 1024: %d" % (
        len(lens), np.median(lens), np.percentile(lens, 90), lens.max(), (lens <= 256).sum(), (lens <= 512).sum(), (lens <= 1024).sum()))
    print("mean wave time %.1f us; sum of wave times / (1024 SIMDs x 6 waves) = %.1f us" % (us.mean(), us.sum() / (1024 * 6)))
    print("totals: rounds %d, iterations %d (full %d), survivors(sum over waves' own chunks) %d" % (t[:, 1].sum(), t[:, 3].sum(), t[:, 4].sum(), (t[:, 2] & 0xFFFFFFFF).sum()))


if __name__ == "__main__":
    main()
