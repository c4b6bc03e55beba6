"""The captured training iteration replayed with a new camera and target every iteration: GPU-bound or host-bound?"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from riggs_amd import synth
from riggs_amd.graph import GraphedTrainStep
from riggs_amd.optim import FusedAdam
w = bench.WORKLOAD
dev, NCAM = "cuda:0", 8
sc, cam0, gm, sw = bench.build_workload(0, dev)
cams = [synth.look_at_camera(w["H"], w["W"], azimuth_deg=45.0 * k, fid=k / NCAM).to(dev) for k in range(NCAM)]
bg = torch.zeros(3, device=dev)
targets = [torch.rand(3, w["H"], w["W"], generator=torch.Generator().manual_seed(k)).to(dev) for k in range(NCAM)]
gm.training_setup(bench._train_args(), capturable=True)
sk = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()], lr=0.0, eps=1e-15, capturable=True)
gts = GraphedTrainStep(gm, sw, cams[0], bg, targets[0], [gm.optimizer, sk], lambda_dssim=0.2, headroom=2.5)
gts.capture(warmup=1)
def loop(n, cyc):
    for it in range(20):
        gts.run(cam=cams[it % NCAM], gt_image=targets[it % NCAM]) if cyc else gts.run()
    torch.cuda.synchronize(); t0 = time.perf_counter(); host = 0.0
    for it in range(n):
        h0 = time.perf_counter()
        gts.run(cam=cams[it % NCAM], gt_image=targets[it % NCAM]) if cyc else gts.run()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, host / n * 1e3
for cyc in (False, True, False, True):
    wall, host = loop(500, cyc)
    print("%s: %.4f ms per iteration, of which the host spends %.4f ms inside run()" % ("cycling cameras + targets" if cyc else "static inputs", wall, host), flush=True)
try:
    gts.check()
except Exception as e:
    print("(check at the end: %s)" % str(e)[:80])
