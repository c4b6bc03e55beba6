"""Kernel times of the fused image loss at 3 x 800 x 800 (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from riggs_amd.loss import l1_ssim
g = torch.Generator().manual_seed(0)
x = torch.rand(3, 800, 800, generator=g).cuda().requires_grad_(True)
y = torch.rand(3, 800, 800, generator=g).cuda()
for _ in range(30):
    l1, s = l1_ssim(x, y)
    (0.8 * l1 + 0.2 * (1 - s)).backward()
    x.grad = None
torch.cuda.synchronize()
