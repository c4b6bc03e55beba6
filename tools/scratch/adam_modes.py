"""The Adam kernel's time by mode: host-side step counts / device-side (capturable) / gated — same tensors (the Gaussians' six)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from riggs_amd import _lib as L
sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
for mode in ("host", "capturable", "gated"):
    gm.training_setup(bench._train_args(), capturable=(mode != "host"))
    if mode == "gated":
        words = torch.zeros(4, dtype=torch.int32, device="cuda")
        gm.optimizer.gate = L.FrameGate([lambda: (words, 0, 0xFFFFFFFF)])
    ps = [g["params"][0] for g in gm.optimizer.param_groups]
    for p in ps:
        p.grad = torch.randn_like(p) * 1e-3
    for _ in range(5):
        gm.optimizer.step()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        gm.optimizer.step()
    b.record(); torch.cuda.synchronize()
    print("%-10s %.1f us per step (events over 50 back-to-back steps)" % (mode, a.elapsed_time(b) / 50 * 1e3))
