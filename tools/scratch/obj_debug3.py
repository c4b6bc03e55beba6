import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_objective import _models
from riggs_amd import synth
from riggs_amd.graph import GraphedTrainStep
J, H, W = 8, 64, 80
sc, gm, sw, opt = _models(1500, J, False)
if sys.argv[1] == "now": sw.use_skinning_weight_mlp = False
if sys.argv[1] == "not": sw.use_template_offsets = False
cam = synth.look_at_camera(H, W, fid=0.3).to("cuda")
bg = torch.zeros(3, device="cuda"); target = torch.rand(3, H, W, device="cuda")
gts = GraphedTrainStep(gm, sw, cam, bg, target, [], lambda_dssim=0.2)
names = {id(p): n for n, p in sw.named_parameters()}
params = list(gts.params)
def snapp():
    torch.cuda.synchronize(); return [None if p.grad is None else p.grad.detach().clone() for p in params]
def diff(tag, A, B):
    bad = [(names.get(id(p), "gm"), float((a-b).abs().max()), float(b.abs().max())) for p, a, b in zip(params, A, B) if a is not None and float((a-b).abs().max()) > 1e-3 * float(b.abs().max())]
    print(tag, bad[:8])
gts._frame(); e1 = snapp()
gts._frame(); e2 = snapp(); diff("eager2 vs eager1", e2, e1)
gts.capture(warmup=1)
def snap():
    torch.cuda.synchronize(); return [None if g is None else g.detach().clone() for g in gts.grads]
gts.run(); g1 = snap(); diff("replay1 vs eager1", g1, e1)
gts.run(); g2 = snap(); diff("replay2 vs eager1", g2, e1)
gts.run(); g3 = snap(); diff("replay3 vs eager1", g3, e1)
