import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_objective import _models
from riggs_amd import synth
from riggs_amd.graph import GraphedTrainStep
from riggs_amd.render import render
from riggs_amd.loss import image_loss
J, H, W = 8, 64, 80
sc, gm, sw, opt = _models(1500, J, False)
cam = synth.look_at_camera(H, W, fid=0.3).to("cuda")
bg = torch.zeros(3, device="cuda"); target = torch.rand(3, H, W, device="cuda")
OPTS = {"both": [gm.optimizer, opt], "gm": [gm.optimizer], "sk": [opt], "no": []}[sys.argv[2]]
gts = GraphedTrainStep(gm, sw, cam, bg, target, OPTS, lambda_dssim=0.2)
gts.capture(warmup=1)
names = {id(p): n for n, p in sw.named_parameters()}
params = list(gts.params)
class Pipe: convert_SHs_python = compute_cov3D_python = debug = False
def snap():
    torch.cuda.synchronize(); return [None if g is None else g.detach().clone() for g in gts.grads]
def diff(tag, A, B):
    bad = [(names.get(id(p), "gm"), float((a-b).abs().max())) for p, a, b in zip(params, A, B) if a is not None and float((a-b).abs().max()) > 1e-3 * float(b.abs().max())]
    print(tag, bad[:6])
gts.run(); g1 = snap()
gts.run(); g2 = snap(); diff("replay2 vs replay1", g2, g1)
mode = sys.argv[1]
if mode == "fwd":      # eager forward only
    with torch.no_grad():
        dv = sw(gm.get_xyz.detach(), sw.expand_time(cam.fid), motion_mask=gm.motion_mask)
elif mode == "bwd":    # eager forward + backward, grads detached from the graph's buffers
    for p in params: p.grad = None
    dv = sw(gm.get_xyz.detach(), sw.expand_time(cam.fid), motion_mask=gm.motion_mask)
    (dv["d_xyz"].sum() + dv["d_rotation"].sum()).backward()
    for p, g in zip(params, gts.grads): p.grad = g
elif mode == "alloc":  # just allocator churn
    xs = [torch.randn(1500, 256, device="cuda") for _ in range(50)]; del xs
torch.cuda.synchronize()
gts.run(); g3 = snap(); diff("after eager %s: replay3 vs replay1" % mode, g3, g1)
