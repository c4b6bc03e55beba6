import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench
from riggs_amd import synth, _lib as L
from riggs_amd.rasterizer import rasterize_forward, RasterArena
from tests import gpu_util as U
w = bench.WORKLOAD
sc = synth.make_surface_scene(w["N"], w["J"], w["seed"]) if len(sys.argv) > 1 else synth.make_scene(w["N"], w["J"], w["seed"])
cam = synth.look_at_camera(w["H"], w["W"])
d = lambda t: t.cuda().contiguous()
st = U.settings_for(cam, [0, 0, 0])
args = (d(sc["xyz"]), d(torch.cat([sc["features_dc"], sc["features_rest"]], 1)), None, d(torch.sigmoid(sc["opacity"])), d(torch.exp(sc["scaling"])), d(torch.nn.functional.normalize(sc["rotation"])), None)
arena = RasterArena()
for k in range(4):
    out = rasterize_forward(st, *args, arena=arena)
    torch.cuda.synchronize()
    b = arena.binning
    stats = b[b.numel() - 256:].view(torch.int32).cpu().numpy()
    print("frame", k, "hist", stats[:24].tolist(), "\n        had", stats[32:56].tolist())
