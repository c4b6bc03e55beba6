"""Case 20 of the fuzz sweep with one knob turned at a time: where does the 3 % on one screen-filling Gaussian come from?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import raster_ref as RR  # noqa: E402
from riggs_amd.rasterizer import rasterize_backward, saved_views  # noqa: E402
from tests import gpu_util as U  # noqa: E402
from tests.test_gpu_fuzz import _case  # noqa: E402

seed = 20
c = _case(seed)
d = lambda t: t.cuda().contiguous()  # noqa: E731


def run(opscale, cot, keep=None, label=""):
    sc, act, cam = U.activated_scene(c["N"], c["J"], 500 + seed, c["H"], c["W"], scale=c["scale"], radius=c["radius"], azimuth_deg=c["azimuth"])
    act["opacities"] = act["opacities"] * opscale
    act["shs"] = act["shs"][:, :4].contiguous()
    if keep is not None:
        act = {k: v[keep].contiguous() for k, v in act.items()}
    out_o, so = U.oracle_forward(act, cam, c["bg"], sh_degree=1)
    H, W = c["H"], c["W"]
    g = torch.Generator().manual_seed(seed)
    if cot == "noise":
        gc = torch.randn(3, H, W, generator=g) / (3 * H * W)
    else:
        gc = torch.ones(3, H, W) / (3 * H * W)
    go = RR.backward(so, gc.numpy(), None, None)
    color, radii, depth, alpha, s = U.hip_forward(act, cam, c["bg"], sh_degree=1)
    gh = rasterize_backward(s, d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]), None, None,
                            None, d(gc), None, None)
    a, b = gh[4].cpu().numpy().reshape(-1), go["opacities"].reshape(-1)
    err = np.abs(a - b) / np.abs(b).max()
    i = int(np.argmax(err))
    print("%-44s worst dL/dopacity err %.2e at g %d (hip %.5e ref %.5e) radius %d, R %d" % (label, err.max(), i, a[i], b[i], int(radii[i]), so.R))


run(0.05, "noise", label="as the case (opacity x0.05, noise cotangent)")
run(0.05, "ones", label="opacity x0.05, constant cotangent")
run(1.0, "noise", label="opacity x1, noise cotangent")
run(0.3, "noise", label="opacity x0.3, noise cotangent")
keep = torch.zeros(c["N"], dtype=torch.bool)
keep[477] = True
run(0.05, "noise", keep=keep, label="Gaussian 477 alone")
keep[:200] = True
run(0.05, "noise", keep=keep, label="Gaussians 0..199 + 477")
