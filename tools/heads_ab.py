"""Does training with the fused 16-bit MLP heads go where training with the reference's fp32 heads goes?

The fused heads' parameter gradients differ from the fp32 arithmetic by 1-6.5 % per tensor (tests/test_gpu_mlp.py).  This fits
one synthetic scene twice from the same seed — fp32 heads (torch Linear = hipBLASLt), fused fp16 heads (csrc/mlp.hip) — for
``iters`` captured training iterations (GraphedTrainStep: PoseMLP, FK, both heads, skinning, render, L1 + SSIM, backward,
FusedAdam of the Gaussians and of the skeleton incl. the heads), cycling over ``n_cams`` cameras / times, against targets a
perturbed copy of the scene rendered (reachable targets: the loss falls), and reports the final loss (mean of the last 100
iterations) and the RMS of the learned deformation d_xyz of both runs.
usage: python tools/heads_ab.py [N] [iters]      -> one JSON line"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.gaussian_model import GaussianModel  # noqa: E402
from riggs_amd.graph import GraphedTrainStep  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402
from riggs_amd.render import render  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
J, H, W, SEED, NCAM = 24, 400, 400, 1236, 8
dev = "cuda:0"


def models(seed_offset=0):
    sc = synth.make_scene(N, J, SEED)
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"], sc["opacity"], device=dev)
    torch.manual_seed(SEED + seed_offset)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).to(dev)
    sw._node_radius.data = sc["node_radius"].to(dev)
    return sc, gm, sw


cams = [synth.look_at_camera(H, W, azimuth_deg=45.0 * k, fid=k / NCAM).to(dev) for k in range(NCAM)]
bg = torch.zeros(3, device=dev)
# targets: the same cloud posed by ANOTHER (seeded) skeleton network with its heads on — what the fitted one has to learn
with torch.no_grad():
    _, gm_t, sw_t = models(seed_offset=1)
    with torch.no_grad():
        sw_t.pose_net.rotation_predictor.weight.mul_(3.0)
        sw_t.detail_net.gaussian_warp.weight.mul_(300.0)
    targets = []
    for c in cams:
        dv = sw_t(gm_t.get_xyz, sw_t.expand_time(c.fid), motion_mask=gm_t.motion_mask)
        targets.append(render(c, gm_t, bench.Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"])["render"].clamp(0, 1).clone())
    del gm_t, sw_t

out = {"what": "same seed, same targets, %d captured training iterations over %d cameras: fp32 heads vs fused fp16 heads" % (ITERS, NCAM),
       "gaussians": N, "image": [H, W], "iterations": ITERS}
curves = {}
for name, fused in (("fp32", False), ("fp32_again", False), ("fused_fp16", True), ("fused_fp16_again", True)):
    sc, gm, sw = models()
    sw.use_fused_heads(fused)
    gm.training_setup(bench._train_args(), capturable=True)
    opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()], lr=0.0, eps=1e-15,
                    capturable=True)
    gts = GraphedTrainStep(gm, sw, cams[0], bg, targets[0], [gm.optimizer, opt], lambda_dssim=0.2, headroom=2.5)
    gts.capture(warmup=1)
    losses = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(ITERS):
        k = it % NCAM
        o = gts.run(cam=cams[k], gt_image=targets[k])
        losses.append(o["loss"].clone())
        if it % 250 == 249:
            gts.check()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / ITERS
    gts.check()
    losses = torch.stack(losses).float().cpu()
    with torch.no_grad():
        rms = []
        for c in cams:
            dv = sw(gm.get_xyz, sw.expand_time(c.fid), motion_mask=gm.motion_mask)
            rms.append(float(dv["d_xyz"].square().mean().sqrt()))
    out[name] = {"ms_per_iteration": round(dt * 1e3, 3), "first_loss": round(float(losses[:NCAM].mean()), 6),
                 "final_loss": round(float(losses[-100:].mean()), 6), "d_xyz_rms": round(sum(rms) / len(rms), 6),
                 "skipped_steps": gts.skipped_steps, "recovered_steps": gts.recovered_steps}
    curves[name] = [round(float(losses[i:i + 100].mean()), 5) for i in range(0, ITERS, max(100, ITERS // 10))]
out["loss_every_tenth"] = curves
# (every run's gradients carry float-atomics noise — the compositing backward sums in a different order each replay — and Adam
# amplifies it: two runs of the SAME arithmetic differ; that spread is the yardstick for the difference between the arithmetics)
rel = lambda x, y: round(abs(x - y) / max(abs(x), 1e-12), 5)  # noqa: E731
a, a2, b, b2 = out["fp32"], out["fp32_again"], out["fused_fp16"], out["fused_fp16_again"]
ma = {k: 0.5 * (a[k] + a2[k]) for k in ("final_loss", "d_xyz_rms")}
mb = {k: 0.5 * (b[k] + b2[k]) for k in ("final_loss", "d_xyz_rms")}
out["final_loss_rel_diff"] = rel(ma["final_loss"], mb["final_loss"])
out["d_xyz_rms_rel_diff"] = rel(ma["d_xyz_rms"], mb["d_xyz_rms"])
out["run_to_run"] = {"fp32_final_loss": rel(a["final_loss"], a2["final_loss"]), "fused_final_loss": rel(b["final_loss"], b2["final_loss"]),
                     "fp32_d_xyz_rms": rel(a["d_xyz_rms"], a2["d_xyz_rms"]), "fused_d_xyz_rms": rel(b["d_xyz_rms"], b2["d_xyz_rms"])}
print(json.dumps(out))
