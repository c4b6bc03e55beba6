"""preprocess_bwd: the one-wave-per-block form against the 256-thread form, by mode (riggs_set_option("preprocess_bwd_lean"))."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd.graph import GraphedFrame  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402

dev = "cuda:0"


def graph_time(gm, sw, cam, sparse, steps=150):
    w = bench.WORKLOAD
    gimg = torch.sign(torch.rand(3, w["H"], w["W"], generator=torch.Generator().manual_seed(3)) - 0.5).to(dev) / (3 * w["H"] * w["W"])
    gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device=dev), bench.params_of(gm, sw), sparse_grad_rows=sparse).capture()
    gf.set_inputs(gimg=gimg)
    for _ in range(10):
        gf.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gf.run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for lean in (0, 1, 0, 1):
    L.set_option("preprocess_bwd_lean", lean)
    sc, cam, gm, sw = bench.build_workload(0, dev)
    a = graph_time(gm, sw, cam, True)
    b = graph_time(gm, sw, cam, False)
    from riggs_amd import synth
    sc2 = synth.make_surface_scene(bench.WORKLOAD["N"], bench.WORKLOAD["J"], 5) if hasattr(synth, "make_surface_scene") else None
    print("lean=%d  headline sparse rows %.4f ms | every row written %.4f ms" % (lean, a, b), flush=True)
L.set_option("preprocess_bwd_lean", -1)
