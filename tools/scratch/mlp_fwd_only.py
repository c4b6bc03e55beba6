"""The fused MLP forward / data-gradient kernels alone, a few launches each, for counter passes:
    rocprofv3 --kernel-trace --pmc <counters> -d out -- python tools/scratch/mlp_fwd_only.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from riggs_amd import mlp as M  # noqa: E402
from riggs_amd.skeleton import WeightMLP  # noqa: E402

N = 300000
torch.manual_seed(0)
x = torch.randn(N, 3, device="cuda") * 0.5
wm = WeightMLP(3, 23).cuda()
xb = M.embed_positions_bf16(x, wm.multires, fmt="fp16")
fh = M.FusedHead(wm.linear, wm.weight_predict, wm.input_ch, wm.skips[0], "fp16")
p = fh._packed()
for _ in range(4):
    M.forward(p, xb[:N], False, xb)
out, (acts, masks) = M.forward(p, xb[:N], True, xb)
g = torch.randn(N, p.out_ch, device="cuda")
sc = M.grad_scale(g)
for _ in range(3):
    M.backward_data(p, g, masks, sc, bias_sums=False)
torch.cuda.synchronize()
