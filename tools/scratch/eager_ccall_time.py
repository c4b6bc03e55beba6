"""Host time of the two C calls of the eagerly issued frame (riggs_frame_forward / riggs_frame_backward) against the whole step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402

lib = L.lib()
acc = {}


def spy(name):
    orig = getattr(lib, name)

    def f(*a):
        t = time.perf_counter()
        r = orig(*a)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(lib, name, f)


for n in ("riggs_frame_forward", "riggs_frame_backward"):
    spy(n)
sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
w = bench.WORKLOAD
gimg = torch.rand(3, w["H"], w["W"], device="cuda") * 1e-6
step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, None, frame_entry=True)
for _ in range(20):
    step()
torch.cuda.synchronize()
acc.clear()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host loop %.1f us/step, with final sync %.1f us/step" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
for k, v in acc.items():
    print("%s: %.1f us per call" % (k, v / n * 1e6))
