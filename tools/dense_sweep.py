"""The opaque-surface ("dense-gradient") scene of bench.py under library options — which forward settings shorten its tail?
usage: python tools/dense_sweep.py [headline]     prints ms per frame per (fwd_wide_min, fwd_wide_tiles, sparse rows) combination,
with the per-kernel event times of one eagerly issued frame (riggs_prof_*) for the first and the best."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.gaussian_model import GaussianModel  # noqa: E402
from riggs_amd.graph import GraphedFrame  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402

dev = "cuda:0"
w = bench.WORKLOAD
headline = len(sys.argv) > 1 and sys.argv[1] == "headline"


def build():
    if headline:
        sc, cam, gm, sw = bench.build_workload(0, dev)
        return gm, sw, cam
    sc = synth.make_surface_scene(w["N"], w["J"], w["seed"])
    cam = synth.look_at_camera(w["H"], w["W"], fid=0.37).to(dev)
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"], sc["opacity"], device=dev)
    torch.manual_seed(w["seed"])
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8, use_skinning_weight_mlp=False,
                      use_template_offsets=False).to(dev)
    sw._node_radius.data = sc["node_radius"].to(dev)
    with torch.no_grad():
        sw.pose_net.rotation_predictor.weight.mul_(0.1)
        sw.pose_net.translation_predictor.weight.mul_(0.1)
    return gm, sw, cam


def run(sparse, steps=60):
    gm, sw, cam = build()
    gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device=dev), bench.params_of(gm, sw), sparse_grad_rows=sparse).capture()
    g = torch.Generator().manual_seed(w["seed"] + 100)
    target = torch.rand(3, w["H"], w["W"], generator=g).to(dev)
    out = gf.run()
    gf.set_inputs(gimg=torch.sign(out["render"].detach() - target) / (3 * w["H"] * w["W"]))
    for _ in range(8):
        gf.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gf.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gf.check()
    return dt * 1e3


extra = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] if len(sys.argv) > 2 else []
combos = extra or [(4096, 256), (2048, 256), (1024, 256), (1024, 1024), (512, 1024), (512, 2048), (256, 2048)]
for sparse in (False, True):
    for wmin, wtiles in combos:
        L.set_option("fwd_wide_min", wmin)
        L.set_option("fwd_wide_tiles", wtiles)
        print("sparse_rows=%d fwd_wide_min=%5d fwd_wide_tiles=%5d : %.4f ms" % (sparse, wmin, wtiles, run(sparse)), flush=True)
