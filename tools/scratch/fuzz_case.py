"""One case of tests/test_gpu_fuzz.py under the microscope: which elements differ, for which Gaussians, in which modes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import raster_ref as RR  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd import rasterizer as RZ  # noqa: E402
from riggs_amd.rasterizer import rasterize_backward, saved_views  # noqa: E402
from tests import gpu_util as U  # noqa: E402
from tests.test_gpu_fuzz import _case  # noqa: E402

seed = int(sys.argv[1])
c = _case(seed)
print(c)
sc, act, cam = U.activated_scene(c["N"], c["J"], 500 + seed, c["H"], c["W"], scale=c["scale"], radius=c["radius"], azimuth_deg=c["azimuth"])
act["opacities"] = act["opacities"] * c["opacity_scale"]
M = (c["deg"] + 1) ** 2
act["shs"] = act["shs"][:, :M].contiguous()
out_o, so = U.oracle_forward(act, cam, c["bg"], sh_degree=c["deg"])
g = torch.Generator().manual_seed(seed)
gc = torch.randn(3, c["H"], c["W"], generator=g) / (3 * c["H"] * c["W"])
gd = torch.randn(1, c["H"], c["W"], generator=g) / (c["H"] * c["W"])
ga = torch.randn(1, c["H"], c["W"], generator=g) / (c["H"] * c["W"])
go = RR.backward(so, gc.numpy(), gd.numpy()[0], ga.numpy()[0])
d = lambda t: t.cuda().contiguous()  # noqa: E731
for ordered in (True, False):
    for grouped in (1, 0):
        RZ.set_ordered_backward(ordered)
        L.set_option("bin_grouped", grouped)
        color, radii, depth, alpha, s = U.hip_forward(act, cam, c["bg"], sh_degree=c["deg"])
        gh = rasterize_backward(s, d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]),
                                d(act["rotations"]), None, None, None, d(gc), d(gd), d(ga))
        a, b = gh[0].cpu().numpy(), go["means3D"]
        err = np.abs(a - b) / np.abs(b).max()
        idx = np.argwhere(err > 1e-4)
        print("ordered", ordered, "grouped", grouped, "bad", len(idx), "max", err.max())
        for i, k in idx[:6]:
            v = saved_views(s)
            print("   g %d comp %d hip %.6e ref %.6e | radius %d depth %.4f tiles %d  opacity %.4f" % (
                i, k, a[i, k], b[i, k], int(radii[i]), float(v["xyd"][i, 2]), int(v["tiles_touched"][i]), float(act["opacities"][i])))
RZ.set_ordered_backward(False)
L.set_option("bin_grouped", -1)
# the oracle in float64? compare the oracle's own sensitivity: perturb nothing, just report the largest |b|
print("max |ref| %.4e" % np.abs(go["means3D"]).max())

# ---- every gradient of the worst Gaussian
RZ.set_ordered_backward(True)
color, radii, depth, alpha, s = U.hip_forward(act, cam, c["bg"], sh_degree=c["deg"])
gh = rasterize_backward(s, d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]),
                        d(act["rotations"]), None, None, None, d(gc), d(gd), d(ga))
RZ.set_ordered_backward(False)
a, b = gh[0].cpu().numpy(), go["means3D"]
i = int(np.argmax(np.abs(a - b).max(1)))
print("worst Gaussian", i, "mean", act["means3D"][i].tolist(), "scale", act["scales"][i].tolist())
v = saved_views(s)
print("  xyd", v["xyd"][i].tolist(), "conic_o", v["conic_o"][i].tolist(), "radius", int(radii[i]))
for got, name in zip(gh, ("means3D", "means2D", "shs", None, "opacities", "scales", "rotations")):
    if name is None:
        continue
    w = go[name]
    print("  %-10s hip %s\n             ref %s" % (name, np.asarray(got.cpu().numpy().reshape(w.shape)[i]).ravel()[:6], np.asarray(w[i]).ravel()[:6]))
