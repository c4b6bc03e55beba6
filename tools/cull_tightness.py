"""How much tighter than the axis-aligned extents would an exact ellipse-vs-block test make the forward's cull?
(bench scene; instances up to each tile's last contributor, 8x4 pixel blocks)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math
import numpy as np, torch
import bench
from riggs_amd.rasterizer import rasterize_forward, saved_views
from tests import gpu_util as U
from riggs_amd import synth

w = bench.WORKLOAD
sc = synth.make_scene(w["N"], w["J"], w["seed"])
cam = synth.look_at_camera(w["H"], w["W"])
d = lambda t: t.cuda().contiguous()
st = U.settings_for(cam, [0, 0, 0])
color, radii, depth, alpha, s = rasterize_forward(
    st, d(sc["xyz"]), d(torch.cat([sc["features_dc"], sc["features_rest"]], 1)), None, d(torch.sigmoid(sc["opacity"])),
    d(torch.exp(sc["scaling"])), d(torch.nn.functional.normalize(sc["rotation"])), None)
v = saved_views(s)
rg = v["ranges"].long(); H, W = w["H"], w["W"]; gx = (W + 15) // 16
nc = torch.zeros(((H + 15) // 16) * 16, gx * 16, dtype=torch.long, device="cuda"); nc[:H, :W] = v["n_contrib"].long()
tmax = nc.reshape(-1, 16, gx, 16).permute(0, 2, 1, 3).reshape(-1, 256).max(1).values
xyd, con, rgb, pl = v["xyd"], v["conic_o"], v["rgb"], v["point_list"].long()
tot_aabb = tot_exact = tot = 0
for t in torch.nonzero(tmax > 0).flatten().tolist():
    lim = int(min(rg[t, 1] - rg[t, 0], tmax[t]))
    ids = pl[rg[t, 0]: rg[t, 0] + lim]
    x, y, hx = xyd[ids, 0], xyd[ids, 1], xyd[ids, 3]
    hy = rgb[ids, 3]
    A, B, Cc, o = con[ids, 0], con[ids, 1], con[ids, 2], con[ids, 3]
    tau = 2.0 * torch.log(255.0 * o)
    tx, ty = (t % gx) * 16, (t // gx) * 16
    for sub in range(8):
        bx0, by0 = tx + (sub & 1) * 8, ty + (sub >> 1) * 4
        bx1, by1 = bx0 + 7, by0 + 3
        aabb = (x + hx >= bx0) & (x - hx <= bx1) & (y + hy >= by0) & (y - hy <= by1)
        # exact: min over the block of q = A dx^2 + 2 B dx dy + C dy^2 (dx = px - x ...): check clamped centre and the edges
        px = x.clamp(bx0, bx1); py = y.clamp(by0, by1)
        best = torch.full_like(x, 1e30)
        def q(px_, py_):
            dx, dy = px_ - x, py_ - y
            return A * dx * dx + 2 * B * dx * dy + Cc * dy * dy
        best = torch.minimum(best, q(px, py))
        for ex in (float(bx0), float(bx1)):   # vertical edges: minimise over y
            dx = ex - x
            yy = (y - B * dx / Cc).clamp(by0, by1)
            best = torch.minimum(best, q(torch.full_like(x, ex), yy))
        for ey in (float(by0), float(by1)):   # horizontal edges
            dy = ey - y
            xx = (x - B * dy / A).clamp(bx0, bx1)
            best = torch.minimum(best, q(xx, torch.full_like(x, ey)))
        exact = aabb & (best <= tau + 0.05)
        tot += lim; tot_aabb += int(aabb.sum()); tot_exact += int(exact.sum())
print("walked (block, instance) pairs %d; survive the extents test %d (%.1f%%); would survive an exact test %d (%.1f%% of those)" % (
    tot, tot_aabb, 100.0 * tot_aabb / tot, tot_exact, 100.0 * tot_exact / max(1, tot_aabb)))
