"""Stage timestamps of the one-launch PoseMLP kernels (workgroup 0), to see where a hop's time goes.
usage: python tools/pose_mlp_trace.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd.skeleton import PoseMLP  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PoseMLP(1, 24 * 4).to(dev)
    bias = torch.tensor([1.0, 0, 0, 0], device=dev)
    t = torch.tensor([0.37], device=dev)
    trace = torch.zeros(128, dtype=torch.int64, device=dev)
    lib = L.lib()
    lib.riggs_pose_mlp_set_trace(trace.data_ptr())
    big = torch.empty(64 << 20, device=dev)
    for it in range(4):
        big.normal_()  # evict the weights from L2 / MALL like the frame's streaming kernels do
        m = net(t, rot_bias=bias)
        (m["rotation"].sum() + m["translation"].sum()).backward()
        torch.cuda.synchronize()
        tr = trace.cpu().numpy()
        f = tr[:64]
        b = tr[64:]
        nf = 2 + 2 * 9
        print("iter", it)
        print("  fwd us since start:", " ".join("%.2f" % ((x - f[0]) / 100.0) for x in f[:nf]))
        nb = 6 + 2 * 8
        print("  bwd us since start:", " ".join("%.2f" % ((x - b[0]) / 100.0) for x in b[:nb]))
    lib.riggs_pose_mlp_set_trace(None)


if __name__ == "__main__":
    main()
