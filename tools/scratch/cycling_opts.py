"""Cycling cameras (a new camera every replay) under the forward work list's history options, on the headline scene and on the
opaque-skin scene: is the previous frame's walk history worth anything when the view jumps?  python tools/scratch/cycling_opts.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from riggs_amd import _lib as L, synth
from riggs_amd.graph import GraphedFrame
w = bench.WORKLOAD
dev = "cuda:0"
def run(surface, n_cams=8, steps=128):
    sc, cam0, gm, sw = bench.build_workload(0, dev, surface=surface)
    cams = [synth.look_at_camera(w["H"], w["W"], azimuth_deg=360.0 * k / n_cams, fid=0.1 + 0.8 * k / n_cams).to(dev) for k in range(n_cams)]
    gf = GraphedFrame(gm, sw, cams[0], torch.zeros(3, device=dev), bench.params_of(gm, sw), sparse_grad_rows=True, headroom=2.5).capture()
    g = torch.Generator().manual_seed(w["seed"] + 7)
    gf.set_inputs(gimg=(torch.sign(torch.rand(3, w["H"], w["W"], generator=g) - 0.5) / (3 * w["H"] * w["W"])).to(dev))
    for k in range(2 * n_cams):
        gf.run(cam=cams[k % n_cams])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        gf.run(cam=cams[k % n_cams])
    torch.cuda.synchronize()
    gf.check()
    return (time.perf_counter() - t0) / steps * 1e3
for surface in (False, True):
    for tol, tiles in ((20, 256), (0, 256), (20, 0), (20, 256), (0, 256)):
        L.set_option("fwd_hist_view_tol", tol); L.set_option("fwd_wide_tiles", tiles)
        print("%s scene, view tolerance %.2f, wide tiles %d: %.4f ms" % ("opaque" if surface else "headline", tol / 100, tiles, run(surface)), flush=True)
