"""Which parameters of synth.make_surface_scene give a dense-gradient frame at the bench sizes?
usage: python tools/dense_scene_probe.py   (prints, per variant: R, visible fraction, fraction of Gaussians with a gradient)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.gaussian_model import GaussianModel  # noqa: E402
from riggs_amd.render import render  # noqa: E402

w = bench.WORKLOAD
cam = synth.look_at_camera(w["H"], w["W"]).to("cuda")
for shell, scale, logit in ((0.09, 0.006, 2.5), (0.15, 0.004, 2.5), (0.2, 0.003, 2.5), (0.25, 0.003, 2.2), (0.2, 0.003, 1.0), (0.3, 0.002, 2.2)):
    sc = synth.make_surface_scene(w["N"], w["J"], w["seed"], shell=shell, scale=scale)
    sc["opacity"] = sc["opacity"] - 2.5 + logit
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"], sc["opacity"], device="cuda")
    pkg = render(cam, gm, bench.Pipe, torch.zeros(3, device="cuda"), 0.0, 0.0, 0.0)
    pkg["render"].backward(torch.full_like(pkg["render"], 1e-6))
    torch.cuda.synchronize()
    print("shell %.2f scale %.3f logit %.1f: visible %.3f with-gradient %.3f alpha-mean %.3f" % (
        shell, scale, logit, float((pkg["radii"] > 0).float().mean()), float((gm._opacity.grad.reshape(-1) != 0).float().mean()),
        float(pkg["alpha"].mean())))
