"""Probe which launches survive hipGraph capture (diagnostic)."""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from riggs_amd import synth
from riggs_amd.rasterizer import rasterize_forward, rasterize_backward, RasterArena
from riggs_amd.skeleton import SkeletonWarp
from tests import gpu_util as U

sc, act, cam = U.activated_scene(20000, 24, 3, 256, 256)
d = lambda t: t.cuda().contiguous()
st = U.settings_for(cam, [0, 0, 0])
args = (d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]), None)
arena = RasterArena()
out = rasterize_forward(st, *args, arena=arena)
torch.cuda.synchronize(); arena.resolve()
gc = torch.ones(3, 256, 256, device="cuda")
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        o = rasterize_forward(st, *args, arena=arena)
        rasterize_backward(o[4], *args, None, None, gc, None, None)
        torch.cuda.current_stream().synchronize(); arena.resolve()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("capturing", what, flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    if what in ("fwd", "both"):
        o = rasterize_forward(st, *args, arena=arena)
    if what in ("bwd", "both"):
        gr = rasterize_backward(o[4], *args, None, None, gc, None, None)
    if what == "deform":
        sw = None
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed ok", float(o[0].sum()))
