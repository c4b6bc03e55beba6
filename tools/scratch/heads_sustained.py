import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from riggs_amd.graph import GraphedFrame, GraphedTrainStep
from riggs_amd.optim import FusedAdam
from riggs_amd.skeleton import SkeletonWarp
dev, w = "cuda:0", bench.WORKLOAD
sc, cam, gm, _ = bench.build_workload(0, dev)
torch.manual_seed(w["seed"])
sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).to(dev).use_fused_heads(True)
sw._node_radius.data = sc["node_radius"].to(dev)
gm.training_setup(bench._train_args(), capturable=True)
opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()], lr=0.0, eps=1e-15, capturable=True)
bg = torch.zeros(3, device=dev)
img0 = GraphedFrame(gm, sw, cam, bg, bench.params_of(gm, sw)).capture().run()["render"].detach().clone()
target = (img0 + 0.05 * torch.randn(img0.shape, generator=torch.Generator().manual_seed(w["seed"] + 7)).to(dev)).clamp_(0.0, 1.0)
for p in gm.parameters() + list(sw.parameters()):
    p.grad = None
gts = GraphedTrainStep(gm, sw, cam, bg, target, [gm.optimizer, opt], lambda_dssim=0.2, sparse_grad_rows=True, lambda_template_offsets=1.0, lambda_template_fixed=100.0)
gts.capture()
for _ in range(5):
    gts.run()
torch.cuda.synchronize()
print("live rows of the WeightMLP's backward at the start: %.4f" % (int(sw._fh_w.last_live_count) / w["N"]))
snap = [p.detach().clone() for p in gm.parameters() + list(sw.parameters())]
ts = []
t_start = time.perf_counter()
while time.perf_counter() - t_start < 6.0:
    t0 = time.perf_counter()
    for _ in range(40):
        gts.run()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 40 * 1e3)
print("blocks of 40 replays over 6 s, ms per replay:", [round(t, 3) for t in ts[:6]], "...", [round(t, 3) for t in ts[-6:]], "n", len(ts))
print("live rows of the WeightMLP's backward now: %.4f; R now: %d" % (int(sw._fh_w.last_live_count) / w["N"], int(gts.check())))
# the same graph replayed WITHOUT training drift: restore the parameters / optimizer state of the start before every block

with torch.no_grad():
    for p, q in zip(gm.parameters() + list(sw.parameters()), snap):
        p.copy_(q)
ts = []
t_start = time.perf_counter()
while time.perf_counter() - t_start < 6.0:
    with torch.no_grad():
        for p, q in zip(gm.parameters() + list(sw.parameters()), snap):
            p.copy_(q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        gts.run()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 40 * 1e3)
print("the same with the parameters put back before every block (no training drift):", [round(t, 3) for t in ts[:4]], "...", [round(t, 3) for t in ts[-4:]], "n", len(ts))
