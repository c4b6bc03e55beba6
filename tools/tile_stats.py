"""Per-tile work distribution of the bench workload (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from riggs_amd.rasterizer import rasterize_forward, saved_views
from tests import gpu_util as U
from riggs_amd import synth

w = bench.WORKLOAD
sc = synth.make_scene(w["N"], w["J"], w["seed"])
cam = synth.look_at_camera(w["H"], w["W"])
d = lambda t: t.cuda().contiguous()
st = U.settings_for(cam, [0, 0, 0])
act = dict(means3D=sc["xyz"], opacities=torch.sigmoid(sc["opacity"]), scales=torch.exp(sc["scaling"]),
           rotations=torch.nn.functional.normalize(sc["rotation"]), shs=torch.cat([sc["features_dc"], sc["features_rest"]], 1))
color, radii, depth, alpha, s = rasterize_forward(st, d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]), None)
v = saved_views(s)
rg = v["ranges"].cpu().numpy().astype(np.int64)
L = rg[:, 1] - rg[:, 0]
nc = v["n_contrib"].cpu().numpy().astype(np.int64)
H, W = w["H"], w["W"]
gx, gy = (W + 15) // 16, (H + 15) // 16
ncp = np.zeros((gy * 16, gx * 16), np.int64); ncp[:H, :W] = nc
tile_max = ncp.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
tile_sum = ncp.reshape(gy, 16, gx, 16).sum(axis=(1, 3)).reshape(-1)
print("R", v["R"], "tiles", len(L), "mean L", L.mean(), "max L", L.max(), "p99", np.percentile(L, 99), "p90", np.percentile(L, 90))
print("sum L*256 (pairs if no early exit)", (L * 256).sum(), " sum processed pairs (sum n_contrib)", tile_sum.sum())
print("tile max n_contrib: mean", tile_max.mean(), "max", tile_max.max(), " sum(max)*256", (tile_max * 256).sum())
print("nonempty tiles", (L > 0).sum())
order = np.sort(L)[::-1]
print("top 10 tile lengths", order[:10])
fT = v["final_T"].cpu().numpy()
print("pixels with T<1e-3:", (fT < 1e-3).mean(), " mean final T", fT.mean())
vis = (radii > 0).sum().item()
print("visible", vis, "tiles/gaussian", v["R"] / vis, "mean radius", radii[radii > 0].float().mean().item())
ne = tile_max[L > 0]
print("tile_max percentiles (non-empty tiles): p10 %d p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(ne, [10, 50, 90, 99, 100])))
print("histogram of tile_max (bins of 256):", np.bincount((ne // 256).astype(np.int64)).tolist())
rowmax = ncp.reshape(gy * 16, gx, 16).max(axis=2)  # per (pixel row, tile column): the chain one forward wave walks
rm = rowmax[rowmax > 0]
print("per-wave chain (row of 16 pixels) percentiles: p50 %d p90 %d p99 %d max %d  n %d" % (*np.percentile(rm, [50, 90, 99, 100]), rm.size))
tt = v["tiles_touched"].cpu().numpy().astype(np.int64)
tt = tt[tt > 0]
print("tiles per visible Gaussian: mean %.1f; <=4: %.2f  <=8: %.2f  <=12: %.2f  <=16: %.2f  <=32: %.2f  max %d" % (
    tt.mean(), (tt <= 4).mean(), (tt <= 8).mean(), (tt <= 12).mean(), (tt <= 16).mean(), (tt <= 32).mean(), tt.max()))
print("share of the instances in Gaussians with <=8 tiles: %.2f, <=16: %.2f" % (tt[tt <= 8].sum() / tt.sum(), tt[tt <= 16].sum() / tt.sum()))
