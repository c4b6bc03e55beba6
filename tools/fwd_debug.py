"""Where does the forward differ from the oracle?  (diagnostic; usage: python tools/fwd_debug.py N J seed H W scale)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riggs_amd.rasterizer import saved_views  # noqa: E402
from tests import gpu_util as U  # noqa: E402

N, J, seed, H, W = (int(x) for x in sys.argv[1:6])
scale = float(sys.argv[6])
sc, act, cam = U.activated_scene(N, J, seed, H, W, scale=scale)
bg = [0.1, 0.3, 0.7]
out_o, so = U.oracle_forward(act, cam, bg)
for rep in range(3):
    color, radii, depth, alpha, s = U.hip_forward(act, cam, bg)
    torch.cuda.synchronize()
    v = saved_views(s)
    nc = v["n_contrib"].cpu().numpy().view(np.uint32)
    rg = v["ranges"].cpu().numpy().astype(np.int64)
    L = rg[:, 1] - rg[:, 0]
    gx = (W + 15) // 16
    bad = np.argwhere(nc != so.n_contrib)
    err = np.abs(color.cpu().numpy() - out_o["color"]).max(0)
    badc = np.argwhere(err > 1e-4)
    print("rep %d: n_contrib differs on %d px, colour on %d px; tiles with > 1024 entries: %d (max list %d)" % (
        rep, len(bad), len(badc), (L > 1024).sum(), L.max()))
    seen = set()
    for (y, x) in list(bad[:2000]) + list(badc[:2000]):
        t = (y // 16) * gx + x // 16
        seen.add(t)
    for t in sorted(seen)[:12]:
        ty, tx = t // gx, t % gx
        blk = (slice(ty * 16, ty * 16 + 16), slice(tx * 16, tx * 16 + 16))
        print("  tile %d len %d: bad n_contrib %d, bad colour %d; oracle n_contrib range [%d, %d]" % (
            t, L[t], (nc[blk] != so.n_contrib[blk]).sum(), (err[blk] > 1e-4).sum(), so.n_contrib[blk].min(), so.n_contrib[blk].max()))
        ys, xs = np.nonzero(nc[blk] != so.n_contrib[blk])
        for y, x in list(zip(ys, xs))[:6]:
            print("     px (%2d,%2d): hip n %d oracle n %d | hip T %.3e oracle T %.3e | colour err %.2e" % (
                y, x, nc[blk][y, x], so.n_contrib[blk][y, x], v["final_T"].cpu().numpy()[blk][y, x], so.final_T[blk][y, x], err[blk][y, x]))
