"""Each of the cycling test's eight cameras replayed on its own (its history is its own view's) against the cycle: what a per-view
history would be worth.  python tools/scratch/per_camera_static.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from riggs_amd import synth
from riggs_amd.graph import GraphedFrame
w = bench.WORKLOAD
dev = "cuda:0"
for surface in (False, True):
    sc, cam0, gm, sw = bench.build_workload(0, dev, surface=surface)
    n_cams = 8
    cams = [synth.look_at_camera(w["H"], w["W"], azimuth_deg=360.0 * k / n_cams, fid=0.1 + 0.8 * k / n_cams).to(dev) for k in range(n_cams)]
    gf = GraphedFrame(gm, sw, cams[0], torch.zeros(3, device=dev), bench.params_of(gm, sw), sparse_grad_rows=True, headroom=2.5).capture()
    g = torch.Generator().manual_seed(w["seed"] + 7)
    gf.set_inputs(gimg=(torch.sign(torch.rand(3, w["H"], w["W"], generator=g) - 0.5) / (3 * w["H"] * w["W"])).to(dev))
    per = []
    for k in range(n_cams):
        gf.set_inputs(cam=cams[k])
        for _ in range(6):
            gf.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(64):
            gf.run()
        torch.cuda.synchronize(); per.append((time.perf_counter() - t0) / 64 * 1e3)
    for k in range(16):
        gf.run(cam=cams[k % n_cams])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(128):
        gf.run(cam=cams[k % n_cams])
    torch.cuda.synchronize(); cyc = (time.perf_counter() - t0) / 128 * 1e3
    gf.check()
    print("%s scene: each camera on its own %s -> mean %.4f ms; cycling %.4f ms" % ("opaque" if surface else "headline", ["%.3f" % v for v in per], sum(per) / len(per), cyc), flush=True)
