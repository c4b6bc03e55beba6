"""Local cost of the gradient-row exchange at the bench workload on ONE GPU: the pack kernel behind a frame, and the ordered
unpack of W segments (W copies of this rank's segment stand in for the peers': same row count per segment as a real run
of similar frames).  Also prints the segment size, i.e. the bytes each xGMI link carries per step and direction.
usage: python tools/exchange_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd.dist import FlatGradAllReduce, SparseRowExchange, row_exchange_order  # noqa: E402
from riggs_amd.graph import GraphedFrame  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    for scene in ("bench", "dense"):
        if scene == "dense":
            from riggs_amd import synth
            from riggs_amd.gaussian_model import GaussianModel
            w = bench.WORKLOAD
            sc = synth.make_surface_scene(w["N"], w["J"], w["seed"])
            _, cam, _, sw = bench.build_workload(0, "cuda:0")
            gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                            sc["opacity"], device="cuda:0")
        else:
            sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
        N = gm.get_xyz.shape[0]
        ordered, n_rows = row_exchange_order(gm, sw)
        bucket = FlatGradAllReduce(ordered)
        gimg = torch.rand(3, cam.image_height, cam.image_width, device="cuda") * 1e-6
        gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device="cuda"), bench.params_of(gm, sw), split_backward=True).capture()
        gf.set_inputs(gimg=gimg)
        gf.run_a()
        gf.run_b()
        rows = [v.view(N, -1) for v in bucket.views[:n_rows]]
        for world in (2, 4, 8):
            ex = SparseRowExchange(rows, capacity=N, world=world)
            ex.pack()
            torch.cuda.synchronize()
            need = int(ex.segment[1])
            ex.resize(int(need * 1.25) + 256)
            t_pack = timed(ex.pack)
            ex.gathered.copy_(ex.segment.repeat(world))
            t_unpack = timed(lambda: ex._unpack(ex))
            assert ex.check()
            gf.run_a()  # (the unpack summed W copies into the gradient buffers: refill them)
            gf.run_b()
            print("%s scene, W=%d: %d of %d rows (%.1f %%), capacity %d, segment %.2f MB (dense bucket rows: %.1f MB); pack %.1f us, "
                  "unpack of %d segments %.1f us" % (scene, world, need, N, 100.0 * need / N, ex.capacity, ex.segment.numel() * 4 / 1e6,
                                                     N * 59 * 4 / 1e6, t_pack, world, t_unpack))
        bucket.unregister()
        del gf


if __name__ == "__main__":
    main()
