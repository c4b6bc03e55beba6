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
    n_items = T * 8 + min(T, L.get_option("fwd_wide_tiles")) * 24  # work items = workgroups of the launch; 4 waves each
    trace = torch.zeros((n_items * 4 * 8 + 4 * 200_000), dtype=torch.int64, device="cuda")
    L.lib().riggs_raster_set_trace_items(n_items)
    gimg = torch.rand(3, w["H"], w["W"], device="cuda") * 1e-6
    step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, FlatGradAllReduce(bench.params_of(gm, sw), register=False))
    step()
    L.lib().riggs_raster_set_trace(trace.data_ptr())
    step()
    torch.cuda.synchronize()
    L.lib().riggs_raster_set_trace(None)
    t = trace.cpu().numpy()[:n_items * 4 * 8].reshape(-1, 8).copy()
    t = t[t[:, 5] != 0]
    tile_of, wide = (t[:, 5] >> 32) & 0xFFFFFF, (t[:, 5] >> 63) & 1
    t[:, 5] &= 0xFFFFFFFF
    us = t[:, 0] / 100.0
    order = np.argsort(-us)
    print("waves with work: %d; time us: max %.1f p99 %.1f p90 %.1f median %.1f" % (len(t), us.max(), *np.percentile(us, [99, 90, 50])))
    print("slowest waves: [us, rounds, survivors/4waves, iterations, full iterations, list length, wide]")
    ph = np.stack([(t[:, 7] >> (16 * k)) & 0xFFFF for k in range(4)], 1) * 64.0  # shader clocks: barrier at the top | stage + stores + barrier | prefetch issue | chunks
    for i in order[:12]:
        print("  %7.1f %5d %7d %6d %6d %7d %d   clocks/round: top %5.0f stage %5.0f issue %5.0f chunks %5.0f" % (
            us[i], t[i, 1], t[i, 2] & 0xFFFFFFFF, t[i, 3], t[i, 4], t[i, 5], wide[i], *(ph[i] / max(1, t[i, 1]))))
    # simple linear model of a wave's time
    A = np.stack([np.ones(len(t)), t[:, 1], t[:, 3] - t[:, 4], t[:, 4]], 1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, us, rcond=None)
    print("fit: us = %.2f + %.3f*rounds + %.3f*skipped_iterations + %.3f*full_iterations" % tuple(coef))
    print("work items that ran: %d (wide %d: %d tiles); tiles %d" % (len(t) // 4, wide.sum() // 4, len(np.unique(tile_of[wide == 1])), len(np.unique(tile_of))))
    for nm, sel in (("8 lanes / pixel", wide == 0), ("32 lanes / pixel", wide == 1)):
        if sel.any():
            r = t[sel, 1] > 0
            print("  %s: us per round (waves with rounds): median %.2f; rounds max %d; wave time max %.1f us" % (
                nm, np.median(us[sel][r] / t[sel, 1][r]), t[sel, 1].max(), us[sel].max()))
    # per tile: list length against the rounds its slowest block walked
    tl = {}
    for k in range(len(t)):
        a = tl.setdefault(int(tile_of[k]), [int(t[k, 5]), 0])
        a[1] = max(a[1], int(t[k, 1]))
    arr = np.array(sorted(tl.values(), reverse=True))
    print("tiles by list length: [length, rounds walked by the slowest block] (longest 24): %s" % " ".join("%d:%d" % (a, b) for a, b in arr[:24]))
    for lo in (1024, 2048, 4096, 8192, 16384):
        sel = arr[:, 0] >= lo
        if sel.any():
            print("  lists >= %5d: %4d tiles; rounds walked: median %d max %d; tiles walking >= 16 rounds: %d" % (
                lo, sel.sum(), np.median(arr[sel, 1]), arr[sel, 1].max(), (arr[sel, 1] >= 16).sum()))
    t0 = t[:, 6].min()
    start, main_end = (t[:, 6] - t0) / 100.0, (t[:, 6] - t0 + t[:, 0]) / 100.0
    print("starts: p50 %.1f p90 %.1f p99 %.1f max %.1f us; compositing loops end: p50 %.1f p99 %.1f max %.1f us" % (
        *np.percentile(start, [50, 90, 99, 100]), *np.percentile(main_end, [50, 99, 100])))
    print("waves in flight (of %d slots) at t us: %s" % (1024 * 6, "  ".join(
        "%d:%d" % (tt, ((start <= tt) & (main_end > tt)).sum()) for tt in range(0, int(main_end.max()) + 10, 10))))
    if (wide == 1).any():
        print("wide blocks: start p50 %.1f max %.1f us, end p50 %.1f max %.1f us" % (
            *np.percentile(start[wide == 1], [50, 100]), *np.percentile(main_end[wide == 1], [50, 100])))
    print("mean wave time %.1f us; sum of wave times / (1024 SIMDs x 6 waves) = %.1f us" % (us.mean(), us.sum() / (1024 * 6)))
    print("totals: rounds %d, iterations %d (full %d), survivors(sum over waves' own chunks) %d" % (t[:, 1].sum(), t[:, 3].sum(), t[:, 4].sum(), (t[:, 2] & 0xFFFFFFFF).sum()))


if __name__ == "__main__":
    main()
