"""ms per iteration (deform + raster fwd + bwd, hipGraph replay) on the configurations of SURVEY.md §8 (C1..C5).
usage: python tools/configs_sweep.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd.graph import GraphedFrame  # noqa: E402

CONFIGS = {"C1": dict(N=10_000, J=8, H=256, W=256), "C2": dict(N=150_000, J=24, H=800, W=800),
           "C3": dict(N=300_000, J=32, H=800, W=800), "C4": dict(N=500_000, J=24, H=1024, W=1024),
           "C5": dict(N=2_000_000, J=64, H=1080, W=1920)}
EXTRA = {"HL": dict(N=300_000, J=24, H=800, W=800)}  # (the bench workload itself, for tools/config_timeline.py / config_counters.sh)


def main():
    for name, cfg in CONFIGS.items():
        bench.WORKLOAD.update(cfg)
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
        params = bench.params_of(gm, sw)
        gimg = torch.rand(3, cfg["H"], cfg["W"], device="cuda") * 1e-6
        gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device="cuda"), params, sparse_grad_rows=True).capture()  # (as bench.py)
        gf.set_inputs(gimg=gimg)
        for _ in range(5):
            gf.run()
        torch.cuda.synchronize()
        R = gf.check()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            gf.run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print("%s N=%d J=%d %dx%d R=%d: %.3f ms/iter = %.0f it/s, %.2f GB" % (name, cfg["N"], cfg["J"], cfg["W"], cfg["H"], R, ms, 1e3 / ms,
                                                                            torch.cuda.max_memory_allocated() / 1e9))
        del gf, gm, sw, sc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
