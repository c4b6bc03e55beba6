"""How many tile instances of the 3-sigma rectangles can never reach alpha >= 1/255 at ANY pixel centre of their tile (dead: they
contribute nothing to the image or to any gradient, whatever the transmittance)?  CPU only: the oracle's forward state of the
bench scene (or `dense`), then per instance (a) the truth by evaluating every pixel centre of the tile, (b) the axis-aligned
alpha-extent box test, (c) the exact ellipse-vs-tile test a kernel could run at emission.   python tools/dead_instances.py [dense]"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import raster_ref as RR  # noqa: E402
from riggs_amd import synth  # noqa: E402
from tests import gpu_util as U  # noqa: E402

w = bench.WORKLOAD
dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
sc = synth.make_surface_scene(w["N"], w["J"], w["seed"]) if dense else synth.make_scene(w["N"], w["J"], w["seed"])
cam = synth.look_at_camera(w["H"], w["W"])
act = dict(means3D=sc["xyz"], opacities=torch.sigmoid(sc["opacity"]), scales=torch.exp(sc["scaling"]),
           rotations=torch.nn.functional.normalize(sc["rotation"]), shs=torch.cat([sc["features_dc"], sc["features_rest"]], 1))
RR.set_threads(8)
out, so = U.oracle_forward(act, cam, [0, 0, 0])
H, W = w["H"], w["W"]
gx = (W + 15) // 16
pl = so.point_list.astype(np.int64)
tile = (so.keys >> np.uint64(32)).astype(np.int64)
R = so.R
xy, co = so.xy[pl], so.conic_o[pl]
A, B, C, o = co[:, 0].astype(np.float64), co[:, 1].astype(np.float64), co[:, 2].astype(np.float64), co[:, 3].astype(np.float64)
tx0, ty0 = (tile % gx) * 16.0, (tile // gx) * 16.0
tx1, ty1 = np.minimum(tx0 + 15, W - 1), np.minimum(ty0 + 15, H - 1)
# (a) truth: any pixel centre of the tile with power <= 0 and o * exp(power) >= 1/255
alive = np.zeros(R, bool)
for a in range(0, R, 200_000):
    b = min(R, a + 200_000)
    px = tx0[a:b, None] + np.arange(16)[None]
    py = ty0[a:b, None] + np.arange(16)[None]
    dx = (xy[a:b, 0:1] - px)[:, None, :]     # (n, 1, 16)
    dy = (xy[a:b, 1:2] - py)[:, :, None]     # (n, 16, 1)
    power = -0.5 * (A[a:b, None, None] * dx * dx + C[a:b, None, None] * dy * dy) - B[a:b, None, None] * dx * dy
    ok = (power <= 0) & (o[a:b, None, None] * np.exp(np.minimum(power, 0)) >= 1.0 / 255.0)
    ok &= (px[:, None, :] < W) & (py[:, :, None] < H)
    alive[a:b] = ok.any(axis=(1, 2))
# (b) alpha-extent box: A dx^2 + 2 B dx dy + C dy^2 <= 2 tau, tau = ln(255 o)
tau = np.log(np.maximum(255.0 * o, 1e-300))
det = np.maximum(A * C - B * B, 1e-30)
never = tau < 0
hx = np.sqrt(np.maximum(2 * tau, 0) * C / det)
hy = np.sqrt(np.maximum(2 * tau, 0) * A / det)
box = (~never) & (xy[:, 0] + hx >= tx0) & (xy[:, 0] - hx <= tx1) & (xy[:, 1] + hy >= ty0) & (xy[:, 1] - hy <= ty1)
# (c) exact: minimum of the quadratic form over the tile's rectangle of pixel centres <= 2 tau
cx, cy = xy[:, 0].astype(np.float64), xy[:, 1].astype(np.float64)
qx = np.clip(cx, tx0, tx1)   # start from the clamped centre, then slide along the faces (2-D convex quadratic over a box)
qy = np.clip(cy, ty0, ty1)


def form(px_, py_):
    dx_, dy_ = px_ - cx, py_ - cy
    return A * dx_ * dx_ + 2 * B * dx_ * dy_ + C * dy_ * dy_
best = form(qx, qy)
# along the vertical face x = qx: optimal y = cy - B (qx - cx) / C, clamped; along the horizontal face y = qy: x = cx - B (qy - cy) / A
y_opt = np.clip(cy - B * (qx - cx) / np.maximum(C, 1e-30), ty0, ty1)
x_opt = np.clip(cx - B * (qy - cy) / np.maximum(A, 1e-30), tx0, tx1)
best = np.minimum(best, np.minimum(form(qx, y_opt), form(x_opt, qy)))
exact = (~never) & (best <= 2 * tau)
print("scene:", "dense" if dense else "headline", " R =", R)
print("alive (truth, pixel centres): %.4f   dead: %.4f" % (alive.mean(), 1 - alive.mean()))
print("kept by the alpha-extent box test: %.4f  (misses alive: %d)" % (box.mean(), int((alive & ~box).sum())))
print("kept by the exact ellipse / tile test: %.4f  (misses alive: %d)" % (exact.mean(), int((alive & ~exact).sum())))
print("Gaussians that can never reach 1/255 (opacity < 1/255): %.4f of the instances" % never.mean())
# where the dead ones sit: per tile list, the share that is dead
L = np.bincount(tile, minlength=gx * ((H + 15) // 16))
D = np.bincount(tile, weights=(~alive).astype(np.float64), minlength=L.size)
big = L >= 4096
print("tiles with >= 4096 instances: %d; their dead share: %.4f" % (big.sum(), D[big].sum() / max(1, L[big].sum())))
