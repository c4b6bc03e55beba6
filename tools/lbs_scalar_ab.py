"""A/B of the all-bones skinning forward: bone records from LDS (the product path) against bone records through the scalar cache
as SGPR operands (riggs_set_option("lbs_scalar", 1 / -1): lbs_forward_scalar_kernel; the library's default picks by size) — same inputs, outputs compared, kernel time from a
graph of 20 launches (so that launch gaps do not enter), at the headline size, at C3 (300 k x 32) and at C5 (2 M x 64).
usage: python tools/lbs_scalar_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.skeleton import fk_forward, lbs_forward  # noqa: E402


def run(N, J, mod=False):
    sc = synth.make_scene(N, J, 7)
    x = sc["xyz"].cuda()
    joints, par = sc["joints"].cuda(), sc["parents"].to(torch.int32).cuda()
    rho = sc["node_radius"].cuda()
    gt = sc["global_trans"].reshape(-1).cuda()
    transforms, node_rot, _ = fk_forward(sc["local_rotation"].cuda().contiguous(), joints, par, gt)
    mask = sc["motion_mask"].reshape(-1).contiguous().cuda()
    wm = torch.rand(N, J - 1, device="cuda") if mod else None
    res = {}
    for flag in (-1, 1):
        L.set_option("lbs_scalar", flag)
        out = lbs_forward(x, joints, par, rho, transforms, node_rot, gt, mask, weight_mod=wm)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(20):
                    lbs_forward(x, joints, par, rho, transforms, node_rot, gt, mask, weight_mod=wm)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
        res[max(flag, 0)] = (best, out[0].clone(), out[1].clone())
    L.set_option("lbs_scalar", 0)
    dx = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max())
    dr = float((res[0][2] - res[1][2]).abs().max() / res[0][2].abs().max())
    print("N=%d J=%d weight_mod=%s: LDS records %.1f us | scalar records %.1f us (incl. its 1-workgroup table launch) | outputs differ by %.1e / %.1e of max"
          % (N, J, mod, res[0][0], res[1][0], dx, dr))


if __name__ == "__main__":
    run(300000, 24)
    run(300000, 32)
    run(300000, 24, mod=True)
    run(2000000, 64)
