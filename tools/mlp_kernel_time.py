"""The fused MLP kernels alone (csrc/mlp.hip, csrc/mlp_wgrad.hip) at the bench size: forward with / without the stores the
backward needs, the data-gradient kernel, the parameter gradients (riggs_mlp_wgrad vs the library GEMMs it replaced), per head.
usage: python tools/mlp_kernel_time.py [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riggs_amd import mlp as M  # noqa: E402
from riggs_amd.skeleton import DeformMLP, WeightMLP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
torch.manual_seed(0)
x = torch.randn(N, 3, device="cuda") * 0.5
wm = WeightMLP(3, 23).cuda()
dn = DeformMLP(xyz_input_ch=3, time_input_ch=96).cuda()
pose = torch.randn(96, device="cuda")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


xb64 = M.embed_positions_bf16(x, wm.multires, fmt="fp16")  # (one operand for both heads: SkeletonWarp._fused_embedding)
for name, net, head, xb, tail in (("WeightMLP", wm, wm.weight_predict, xb64, None),
                                  ("DeformMLP, the pose through its biases (what SkeletonWarp runs)", dn, dn.gaussian_warp, xb64, pose),
                                  ("DeformMLP, the pose as 96 operand columns of every row", dn, dn.gaussian_warp,
                                   M.embed_positions_bf16(x, dn.multires, pose, fmt="fp16"), None)):
    n_tail = 0 if tail is None else tail.numel()
    fh = M.FusedHead(net.linear, head, net.input_ch - n_tail, net.skips[0], "fp16", tail_ch=n_tail)
    p = fh._packed()
    if n_tail:
        p.set_tail(tail)
    # (the flops the MLP is defined by: the tail's columns are flops the fold does not spend)
    flops = 2.0 * N * sum(q.numel() for n_, q in net.named_parameters() if q.dim() == 2)
    t_inf = timed(lambda: M.forward(p, xb[:N], False, xb))
    t_fwd = timed(lambda: M.forward(p, xb[:N], True, xb))
    out, (acts, masks) = M.forward(p, xb[:N], True, xb)
    g = torch.randn(N, p.out_ch, device="cuda")
    sc = M.grad_scale(g)
    t_bwd = timed(lambda: M.backward_data(p, g, masks, sc, bias_sums=False))
    t_bwd_b = timed(lambda: M.backward_data(p, g, masks, sc))
    dpre, db = M.backward_data(p, g, masks, sc)
    t_wg = timed(lambda: M.param_grads(p, xb, acts, dpre, g, sc, tail=tail))
    t_lib = timed(lambda: M.library_param_grads(p, xb, acts, dpre, db, g, sc)) if not n_tail else float("nan")
    gb = (2 * (p.depth - 1) * N * 512 + 2 * N * (512 + 2 * p.in_pad) + N * (512 + 64)) / 1e9
    print("%s N=%d: forward %.3f ms (%.0f TFLOP/s; %.3f ms without the activation / mask stores), data gradient %.3f ms (%.0f TFLOP/s)"
          % (name, N, t_fwd, flops / t_fwd / 1e9, t_inf, t_bwd, flops / t_bwd / 1e9))
    print("    parameter gradients: riggs_mlp_wgrad %.3f ms (%.2f GB of operands: %.2f TB/s) | library GEMMs %.3f ms + bias sums in the "
          "data-gradient kernel %.3f ms" % (t_wg, gb, gb / t_wg, t_lib, t_bwd_b - t_bwd))
