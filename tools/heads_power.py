"""Is the heads-on captured training iteration bound by the chip's power budget?  Replays it (and, for contrast, the headline frame)
for a few seconds each while a thread samples rocm-smi: shader clock, socket power, and the replay rate.
usage: python tools/heads_power.py"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd.graph import GraphedFrame, GraphedTrainStep  # noqa: E402
from riggs_amd.optim import FusedAdam  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(d.keys())[0]]
            sclk = [v for k, v in card.items() if "sclk" in k.lower()]
            pw = [v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()]
            out.append((sclk[0] if sclk else None, pw[0] if pw else None))
        except Exception as e:  # noqa: BLE001
            out.append(("err", str(e)[:60]))
        time.sleep(0.15)


def measure(label, run, seconds=4.0):
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    print("%s: %.4f ms per replay; rocm-smi samples (sclk, W): %s" % (label, dt / n * 1e3, out[2:12]), flush=True)


def main():
    dev, w = "cuda:0", bench.WORKLOAD
    sc, cam, gm, sw0 = bench.build_workload(0, dev)
    gf = GraphedFrame(gm, sw0, cam, torch.zeros(3, device=dev), bench.params_of(gm, sw0)).capture()
    measure("headline frame (no heads)", gf.run)
    del gf
    sc, cam, gm, _ = bench.build_workload(0, dev)
    torch.manual_seed(w["seed"])
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).to(dev).use_fused_heads(True)
    sw._node_radius.data = sc["node_radius"].to(dev)
    gm.training_setup(bench._train_args(), capturable=True)
    opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()], lr=0.0, eps=1e-15,
                    capturable=True)
    bg = torch.zeros(3, device=dev)
    img0 = GraphedFrame(gm, sw, cam, bg, bench.params_of(gm, sw)).capture().run()["render"].detach().clone()
    target = (img0 + 0.05 * torch.randn(img0.shape, generator=torch.Generator().manual_seed(w["seed"] + 7)).to(dev)).clamp_(0.0, 1.0)
    for p in gm.parameters() + list(sw.parameters()):
        p.grad = None
    gts = GraphedTrainStep(gm, sw, cam, bg, target, [gm.optimizer, opt], lambda_dssim=0.2, sparse_grad_rows=True,
                           lambda_template_offsets=1.0, lambda_template_fixed=100.0)
    gts.capture()
    measure("heads-on training iteration", gts.run)
    # the same iteration with an idle gap behind every replay: if the chip is power-bound, the busy part gets faster
    for gap_ms in (1.0, 3.0):
        def run_gap():
            gts.run()
            torch.cuda.synchronize()
            time.sleep(gap_ms * 1e-3)
        for _ in range(20):
            run_gap()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for _ in range(200):
            ev0.record(); gts.run(); ev1.record()
            torch.cuda.synchronize()
            tot += ev0.elapsed_time(ev1)
            time.sleep(gap_ms * 1e-3)
        print("heads-on iteration with %.0f ms of idle behind every replay: %.4f ms per replay (events)" % (gap_ms, tot / 200), flush=True)


if __name__ == "__main__":
    main()
