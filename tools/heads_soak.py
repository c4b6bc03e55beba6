"""A long run of the heads-on captured training iteration (both fused MLP heads, the stage-2 objective, both FusedAdam steps: one
hipGraph) over cycling cameras and targets: is every parameter and moment finite at the end, did the gate skip a step, what did the
losses do, did the iteration's time drift.  The fused heads' kernels wait for their weight fragments with hand-counted s_waitcnt
vmcnt beside stores in flight (csrc/mlp.hip): a wrong count would show here as garbage sooner or later.
usage: python tools/heads_soak.py [iterations]   -> one JSON line"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.graph import GraphedFrame, GraphedTrainStep  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
w = bench.WORKLOAD
dev, NCAM = "cuda:0", 8
sc, cam0, gm, _sw = bench.build_workload(0, dev)
torch.manual_seed(w["seed"])
sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8, use_skinning_weight_mlp=True,
                  use_template_offsets=True).to(dev)
sw._node_radius.data = sc["node_radius"].to(dev)
sw.use_fused_heads(True)
cams = [synth.look_at_camera(w["H"], w["W"], azimuth_deg=45.0 * k, fid=k / NCAM).to(dev) for k in range(NCAM)]
bg = torch.zeros(3, device=dev)
gm.training_setup(bench._train_args(), capturable=True)
sk = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()], lr=0.0, eps=1e-15, capturable=True)
targets = []
for k, c in enumerate(cams):  # targets: each camera's own first render plus noise (reachable)
    img = GraphedFrame(gm, sw, c, bg, bench.params_of(gm, sw)).capture().run()["render"].detach().clone()
    targets.append((img + 0.05 * torch.randn(img.shape, generator=torch.Generator().manual_seed(w["seed"] + k)).to(dev)).clamp_(0.0, 1.0))
for p in gm.parameters() + list(sw.parameters()):
    p.grad = None
gts = GraphedTrainStep(gm, sw, cams[0], bg, targets[0], [gm.optimizer, sk], lambda_dssim=0.2, headroom=2.5, sparse_grad_rows=True,
                       lambda_template_offsets=1.0, lambda_template_fixed=100.0)
gts.capture(warmup=1)
first = None
skipped = recovered = 0
times = []
torch.cuda.synchronize()
t0 = t_blk = time.perf_counter()
for it in range(ITERS):
    k = it % NCAM
    out = gts.run(cam=cams[k], gt_image=targets[k], is_template=(k == 0))
    if it == NCAM - 1:
        first = (float(out["loss"]), float(out["template_offsets_loss"]))
    if it % 1000 == 999:
        gts.check()
        skipped += gts.skipped_steps
        recovered += gts.recovered_steps
        torch.cuda.synchronize()
        now = time.perf_counter()
        times.append(round((now - t_blk), 3))
        t_blk = now
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gts.check()
last = (float(out["loss"]), float(out["template_offsets_loss"]))
params = [p for o in (gm.optimizer, sk) for g in o.param_groups for p in g["params"]]
finite = all(bool(torch.isfinite(p).all()) for p in params) and all(
    bool(torch.isfinite(st["exp_avg"]).all() and torch.isfinite(st["exp_avg_sq"]).all()) for o in (gm.optimizer, sk) for st in o.state.values())
print(json.dumps({"what": "heads-on captured training iterations over %d cameras (the template frame every %d-th), check() every 1000" % (NCAM, NCAM),
                  "iterations": ITERS, "gaussians": w["N"], "ms_per_iteration": round(dt / ITERS * 1e3, 4),
                  "seconds_per_1000_iterations": times, "loss_and_template_loss_first": first, "loss_and_template_loss_last": last,
                  "skipped_steps": int(skipped), "recovered_steps": int(recovered), "everything_finite": bool(finite),
                  "weight_mlp_live_rows_last": int(sw._fh_w.last_live_count) / float(w["N"])}))
