"""Time the LBS backward with a dense and with a sparse incoming gradient (the library's own HIP-event timers).
usage: python tools/lbs_bwd_bench.py [fraction_with_gradient ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import bench  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402


def main(fracs):
    sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    x = gm._xyz.detach().clone().requires_grad_(True)
    J = sw.joints.shape[0] if hasattr(sw, "joints") else bench.WORKLOAD["J"]
    q = torch.nn.functional.normalize(torch.randn(J, 4, device="cuda"), dim=-1).requires_grad_(True)
    gt = torch.zeros(3, device="cuda", requires_grad=True)
    N = x.shape[0]
    for f in fracs:
        keep = (torch.rand(N, 1, device="cuda") < f).float()
        gx = torch.randn(N, 3, device="cuda") * keep
        gr = torch.randn(N, 4, device="cuda") * keep
        lib = L.lib()
        names = [lib.riggs_prof_name(k).decode() for k in range(lib.riggs_prof_count())]
        lib.riggs_prof_reset()
        lib.riggs_prof_enable(0xFFFFFFFF)
        for it in range(12):
            out = sw.deform_by_pose(x, {"local_rotation": q, "global_trans": gt}, None)
            torch.autograd.backward((out["d_xyz"], out["d_rotation"]), (gx, gr))
            torch.cuda.synchronize()
            x.grad = None; q.grad = None; gt.grad = None
        lib.riggs_prof_enable(0)
        res = {}
        for k, nm in enumerate(names):
            tot, cnt = C.c_float(0), C.c_int32(0)
            lib.riggs_prof_read(k, C.byref(tot), C.byref(cnt))
            if cnt.value and "lbs" in nm:
                res[nm] = round(1e3 * tot.value / cnt.value, 1)
        print("fraction of Gaussians with a gradient %.2f: us per launch" % f, res)


if __name__ == "__main__":
    main([float(v) for v in sys.argv[1:]] or [1.0, 0.3, 0.07])
