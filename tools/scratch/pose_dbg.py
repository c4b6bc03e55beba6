import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import bench
from riggs_amd.rasterizer import RasterArena
from riggs_amd.graph import GraphedFrame
bench.WORKLOAD.update(N=20000, J=24, H=128, W=128)
sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
w = bench.WORKLOAD
gimg = (torch.sign(torch.rand(3, w["H"], w["W"], generator=torch.Generator().manual_seed(1)) - 0.5) / (3 * w["H"] * w["W"])).cuda()
params = bench.params_of(gm, sw)
for mode in ("eager", "graph"):
    if mode == "eager":
        step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, None)
    else:
        for p in params: p.grad = None
        gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device="cuda"), params, sparse_grad_rows=True).capture()
        gf.set_inputs(gimg=gimg)
        step = gf.run
    step(); step()
    torch.cuda.synchronize()
    hp = [p.grad.detach().clone() for p in sw.pose_net.parameters()]
    with torch.no_grad():
        t_in = sw.expand_time(cam.fid); na = sw.get_pose_info(t_in); dv = sw(gm.get_xyz.detach(), t_in, motion_mask=gm.motion_mask)
    pose = (na["local_rotation"].cpu(), na["global_trans"].cpu())
    _, og, _ = bench._oracle_iteration(sc, cam.to("cpu"), gimg.cpu(), pose, deformed=(dv["d_xyz"].cpu(), dv["d_rotation"].cpu()))
    net = copy.deepcopy(sw).cpu().double()
    for p in net.pose_net.parameters(): p.grad = None
    na2 = net.get_pose_info(net.expand_time(cam.fid.detach().cpu().double()))
    torch.autograd.backward([na2["local_rotation"], na2["global_trans"]], [torch.from_numpy(og["local_rotation"]).double(), torch.from_numpy(og["global_trans"]).double()])
    ref = [p.grad for p in net.pose_net.parameters()]
    tot = max(float(g.abs().max()) for g in ref)
    print(mode, "tot", tot, [round(float((g - h.double().cpu()).abs().max()) / tot, 6) for g, h in zip(ref, hp)][:20])
    print("   hip norms", [round(float(h.abs().max()), 9) for h in hp][:6], "ref", [round(float(g.abs().max()), 9) for g in ref][:6])
