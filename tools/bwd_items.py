"""Work items of the compositing backward on the bench workload: how many chunks, how much of each is live."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
rg = v["ranges"].cpu().numpy().astype(np.int64)
L = rg[:, 1] - rg[:, 0]
nc = v["n_contrib"].cpu().numpy().astype(np.int64)
H, W = w["H"], w["W"]
gx, gy = (W + 15) // 16, (H + 15) // 16
ncp = np.zeros((gy * 16, gx * 16), np.int64); ncp[:H, :W] = nc
per_tile = ncp.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(gy * gx, 256)
tmax = per_tile.max(1)
lim = np.minimum(L[:gx * gy], tmax)
chunks = (lim + 63) // 64
print("tiles with work %d, chunks (work items) %d, instances walked %d of R=%d" % ((chunks > 0).sum(), chunks.sum(), lim.sum(), L.sum()))
# live pixels per chunk: pixels whose n_contrib > 64 * chunk
live = []
for t in np.nonzero(chunks)[0]:
    n = np.sort(per_tile[t])
    for c in range(chunks[t]):
        live.append(256 - np.searchsorted(n, 64 * c, side="right"))
live = np.array(live)
print("live pixels per chunk: mean %.1f, p10 %d p50 %d p90 %d; chunks with <=64 live pixels: %.1f%%" % (live.mean(), *np.percentile(live, [10, 50, 90]), 100 * (live <= 64).mean()))
print("(instance, pixel) pairs offered to the arithmetic: %.1f M; pairs with alpha >= 1/255 (sum n_contrib upper bound): %.1f M" % (live.sum() * 64 / 1e6, nc.sum() / 1e6))
