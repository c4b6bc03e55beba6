import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_gpu_objective import _models
from riggs_amd import synth
from riggs_amd.graph import GraphedTrainStep
from riggs_amd.loss import image_loss
from riggs_amd.render import render
fused, N = False, 1500
LT = None if sys.argv[1] == "n" else 1.0
LF = None if sys.argv[2] == "n" else 100.0
J, H, W = 8, 64, 80
sc, gm, sw, opt = _models(N, J, fused)
cam = synth.look_at_camera(H, W, fid=0.3).to("cuda")
bg = torch.zeros(3, device="cuda"); target = torch.rand(3, H, W, device="cuda")
gts = GraphedTrainStep(gm, sw, cam, bg, target, [gm.optimizer, opt], lambda_dssim=0.2, lambda_template_offsets=LT, lambda_template_fixed=LF)
gts.capture(warmup=1)
names = {id(p): n for n, p in sw.named_parameters()}
names.update({id(p): "gm%d" % i for i, p in enumerate(gm.parameters())})
params = list(gts.params)
unit = torch.tensor([1.0, 0, 0, 0], device="cuda")
class Pipe: convert_SHs_python = compute_cov3D_python = debug = False
for is_t in (False, True, False):
    gts.run(is_template=is_t); torch.cuda.synchronize()
    got = [None if g is None else g.detach().clone() for g in gts.grads]
    for p in params: p.grad = None
    dv = sw(gm.get_xyz.detach(), sw.expand_time(cam.fid), motion_mask=gm.motion_mask)
    pkg = render(cam, gm, Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"])
    loss, _ = image_loss(pkg["render"], target, 0.2)
    t_loss = (sw.template_offsets ** 2).mean()
    f_loss = ((dv["local_rotation"].reshape(-1, 4) - unit) ** 2).mean()
    total = loss + (0 if LT is None else 1.0 * (1e3 if is_t else 1.0) * t_loss) + (100.0 * f_loss if (is_t and LF) else 0.0)
    total.backward(); torch.cuda.synchronize()
    print("is_t", is_t, sys.argv[1:])
    for p, a in zip(params, got):
        b = p.grad
        if b is None or a is None:
            print(names.get(id(p)), "None", a is None, b is None); continue
        if float((a-b).abs().max()) > 1e-3 * float(b.abs().max()): print("%-40s a %.3e b %.3e err %.3e" % (names.get(id(p)), float(a.abs().max()), float(b.abs().max()), float((a-b).abs().max())))
    for p, g in zip(params, gts.grads): p.grad = g
