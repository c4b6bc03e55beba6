"""Host time of the eager frame entry by section: Python before / inside / after the C calls, autograd's own share."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from riggs_amd import frame as F  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402

acc = {}
pc = time.perf_counter


def wrap(cls, name, key):
    orig = getattr(cls, name)

    def f(*a, **k):
        t = pc()
        r = orig(*a, **k)
        acc[key] = acc.get(key, 0.0) + pc() - t
        return r
    setattr(cls, name, staticmethod(f))


wrap(F._FrameFn, "forward", "FrameFn.forward (python + C)")
wrap(F._FrameFn, "backward", "FrameFn.backward (python + C)")
sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
w = bench.WORKLOAD
gimg = torch.rand(3, w["H"], w["W"], device="cuda") * 1e-6
bg = torch.zeros(3, device="cuda")
params = bench.params_of(gm, sw)
arena = RasterArena()


def step():
    t = pc()
    for p in params:
        p.grad = None
    acc["grad=None"] = acc.get("grad=None", 0.0) + pc() - t
    t = pc()
    pkg = F.deform_render(cam, gm, sw, bench.Pipe, bg, arena=arena)
    acc["deform_render total"] = acc.get("deform_render total", 0.0) + pc() - t
    t = pc()
    pkg["render"].backward(gimg)
    acc["backward total"] = acc.get("backward total", 0.0) + pc() - t


for _ in range(20):
    step()
torch.cuda.synchronize()
acc.clear()
n = 300
t0 = pc()
for _ in range(n):
    step()
t1 = pc()
torch.cuda.synchronize()
print("host loop %.1f us/step" % ((t1 - t0) / n * 1e6))
for k, v in acc.items():
    print("  %-36s %.1f us" % (k, v / n * 1e6))
for k, v in sorted(getattr(F, '_ACC', {}).items()):
    print("  %-36s %.1f us" % (k, v / (n + 0) * 1e6))
