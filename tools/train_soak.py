"""A long run of the captured training iteration (GraphedTrainStep: deform -> render -> L1 + SSIM -> backward -> both FusedAdam steps,
one hipGraph) over cycling cameras and targets: how many replays were skipped behind the frame's valid gate, how many were repaired,
whether anything went non-finite, what the loss did.  usage: python tools/train_soak.py [iterations] [N]   -> one JSON line"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.graph import GraphedTrainStep  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402
from riggs_amd.render import render  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
if len(sys.argv) > 2:
    bench.WORKLOAD.update(N=int(sys.argv[2]))
w = bench.WORKLOAD
dev, NCAM = "cuda:0", 8
sc, cam0, gm, sw = bench.build_workload(0, dev)
cams = [synth.look_at_camera(w["H"], w["W"], azimuth_deg=45.0 * k, fid=k / NCAM).to(dev) for k in range(NCAM)]
bg = torch.zeros(3, device=dev)
with torch.no_grad():  # targets: the scene itself under a perturbed skeleton network (reachable: the loss falls)
    sc2, _, gm2, sw2 = bench.build_workload(0, dev)
    sw2.pose_net.rotation_predictor.weight.mul_(2.0)
    targets = []
    for c in cams:
        dv = sw2(gm2.get_xyz, sw2.expand_time(c.fid), motion_mask=gm2.motion_mask)
        targets.append(render(c, gm2, bench.Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"])["render"].clamp(0, 1).clone())
    del gm2, sw2
gm.training_setup(bench._train_args(), capturable=True)
sk = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()], lr=0.0, eps=1e-15, capturable=True)
gts = GraphedTrainStep(gm, sw, cams[0], bg, targets[0], [gm.optimizer, sk], lambda_dssim=0.2, headroom=2.5)
gts.capture(warmup=1)
first = last = None
skipped = recovered = 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(ITERS):
    k = it % NCAM
    out = gts.run(cam=cams[k], gt_image=targets[k])
    if it == NCAM - 1:
        first = float(out["loss"])
    if it % 1000 == 999:
        gts.check()
        skipped += gts.skipped_steps
        recovered += gts.recovered_steps
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gts.check()
skipped += gts.skipped_steps
recovered += gts.recovered_steps
last = float(out["loss"])
params = [p for o in (gm.optimizer, sk) for g in o.param_groups for p in g["params"]]
finite = all(bool(torch.isfinite(p).all()) for p in params) and all(
    bool(torch.isfinite(st["exp_avg"]).all() and torch.isfinite(st["exp_avg_sq"]).all()) for o in (gm.optimizer, sk) for st in o.state.values())
print(json.dumps({"what": "captured training iterations over %d cameras, check() every 1000" % NCAM, "iterations": ITERS, "gaussians": w["N"],
                  "image": [w["H"], w["W"]], "ms_per_iteration": round(dt / ITERS * 1e3, 4), "skipped_steps": skipped,
                  "recovered_steps": recovered, "all_parameters_and_moments_finite": finite, "loss_after_8": first, "loss_at_end": last,
                  "adam_steps": float(gm.optimizer.state[gm._xyz]["step"])}))
