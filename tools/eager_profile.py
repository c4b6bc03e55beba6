"""Where does the host time of the EAGER drop-in frame go (what an unmodified train_rig.py:535-554 issues: skeleton forward ->
render() -> backward, every launch through ctypes + autograd)?  python tools/eager_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd.dist import FlatGradAllReduce  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    w = bench.WORKLOAD
    gimg = torch.rand(3, w["H"], w["W"], device="cuda") * 1e-6
    for label, bucket in (("fresh gradient buffers", None), ("flat gradient bucket", FlatGradAllReduce(bench.params_of(gm, sw)))):
        step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, bucket)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        print("%s: %.4f ms per eager frame" % (label, (time.perf_counter() - t0) / steps * 1e3))
        if bucket is not None:
            bucket.unregister()
    fe = len(sys.argv) > 2 and sys.argv[2] == "frame"
    step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, None, frame_entry=fe)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
