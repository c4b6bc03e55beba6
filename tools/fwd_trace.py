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
    n_items = (T + 8_000_000 // 1024) * 8  # work items: 8 pixel blocks per (tile, segment of 1024 instances); 4 waves each
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
    tile_of, seg_of, local = (t[:, 5] >> 32) & 0xFFFF, (t[:, 5] >> 48) & 0x7FFF, (t[:, 5] >> 63) & 1
    t[:, 5] &= 0xFFFFFFFF
    us = t[:, 0] / 100.0
    order = np.argsort(-us)
    print("waves with work: %d; time us: max %.1f p99 %.1f p90 %.1f median %.1f" % (len(t), us.max(), *np.percentile(us, [99, 90, 50])))
    print("slowest waves: [us, rounds, survivors/4waves, iterations, full iterations, segment length]")
    for i in order[:12]:
        print("  %7.1f %5d %7d %6d %6d %7d" % (us[i], t[i, 1], t[i, 2] & 0xFFFFFFFF, t[i, 3], t[i, 4], t[i, 5]))
    # simple linear model of a wave's time
    A = np.stack([np.ones(len(t)), t[:, 1], t[:, 3] - t[:, 4], t[:, 4]], 1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, us, rcond=None)
    print("fit: us = %.2f + %.3f*rounds + %.3f*skipped_iterations + %.3f*full_iterations" % tuple(coef))
    print("work items that ran: %d (first segments %d, deeper %d); tiles %d; deepest segment %d" % (
        len(t) // 4, (seg_of == 0).sum() // 4, (seg_of > 0).sum() // 4, len(np.unique(tile_of)), seg_of.max()))
    deep = seg_of > 0
    print("deeper segments that ran as continuations: %d, from T = 1 (local): %d" % (((deep) & (local == 0)).sum() // 4, (local == 1).sum() // 4))
    if deep.any():
        print("deeper segments: rounds walked %d of %d possible (%.0f %% — the rest was cut short by dead_from)" % (
            t[deep, 1].sum(), (np.ceil(t[deep, 5] / 256)).sum(), 100.0 * t[deep, 1].sum() / np.ceil(t[deep, 5] / 256).sum()))
    t0 = t[:, 6].min()
    start, main_end = (t[:, 6] - t0) / 100.0, (t[:, 6] - t0 + t[:, 0]) / 100.0
    chain_us, steps, walk, back = (t[:, 7] & 0xFFFFFFFF) / 100.0, (t[:, 7] >> 32) & 0xFFF, (t[:, 7] >> 44) & 0x3FF, (t[:, 7] >> 54) & 0x3FF
    print("starts: p50 %.1f p90 %.1f p99 %.1f max %.1f us; compositing loops end: p50 %.1f p99 %.1f max %.1f us" % (
        *np.percentile(start, [50, 90, 99, 100]), *np.percentile(main_end, [50, 99, 100])))
    endall = main_end + chain_us
    lvl0 = seg_of == 0
    print("starts of first segments: p50 %.1f p99 %.1f max %.1f; of deeper ones: p1 %.1f p50 %.1f max %.1f" % (
        *np.percentile(start[lvl0], [50, 99, 100]), *(np.percentile(start[~lvl0], [1, 50, 100]) if (~lvl0).any() else (0, 0, 0))))
    print("waves in flight (of %d slots) at t us: %s" % (1024 * 6, "  ".join(
        "%d:%d" % (tt, ((start <= tt) & (endall > tt)).sum()) for tt in range(0, int(endall.max()) + 10, 10))))
    ch = chain_us > 0
    if ch.any():
        end = main_end + chain_us
        print("chains: %d waves took part; time in the chain p50 %.1f p90 %.1f p99 %.1f max %.1f us; last chain ends at %.1f us" % (
            ch.sum(), *np.percentile(chain_us[ch], [50, 90, 99, 100]), end.max()))
        print("        segments combined: total %d, max per holder %d; rounds composited (again, or of segments walked here): total %d, max %d; loop passes | own walks << 5: total %d, max %d" % (
            steps.sum() // 1, steps.max(), walk.sum(), walk.max(), back.sum(), back.max()))
        o2 = np.argsort(-chain_us)[:10]
        print("        slowest chain holders: [chain us, segments combined, rounds composited, passes | own walks << 5, tile, seg, start us]")
        for k in o2:
            print("          %7.1f %4d %4d %4d %6d %4d %7.1f" % (chain_us[k], steps[k], walk[k], back[k], tile_of[k], seg_of[k], start[k]))
    print("mean wave time %.1f us; sum of wave times / (1024 SIMDs x 6 waves) = %.1f us" % (us.mean(), us.sum() / (1024 * 6)))
    print("totals: rounds %d, iterations %d (full %d), survivors(sum over waves' own chunks) %d" % (t[:, 1].sum(), t[:, 3].sum(), t[:, 4].sum(), (t[:, 2] & 0xFFFFFFFF).sum()))


if __name__ == "__main__":
    main()
