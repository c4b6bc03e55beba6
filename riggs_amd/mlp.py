"""Fused per-Gaussian MLP heads on the CDNA4 matrix cores (csrc/mlp.hip).

``FusedHead`` wraps a WeightMLP / DeformMLP host mirror (riggs_amd.skeleton): same parameters (fp32 masters, the ones
the optimizer and the checkpoints see), but forward and backward run as ONE HIP launch each with bf16 operands and
fp32 accumulation; the weight gradients are (256 x N)·(N x K) GEMMs handed to hipBLASLt (bf16 in, fp32 out).
The reference computes these MLPs in fp32 (skeleton_utils/network_utils.py:6-112), so this path is OPT-IN
(``SkeletonWarp.use_fused_heads(True)``) and is tested against the fp32 mirrors at bf16 tolerance.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

ROWS = 64


def layout_probe() -> torch.Tensor:
    """D = A·B from the kernel's self-test (A = [I_16; 0], asymmetric B): rows 0..15 must equal B."""
    out = torch.zeros(32, 32, device="cuda")
    L.check(L.lib().riggs_mlp_layout_probe(out.data_ptr(), L.stream_ptr()), "riggs_mlp_layout_probe")
    return out


def expected_probe() -> torch.Tensor:
    k = torch.arange(16, dtype=torch.float32)[:, None]
    n = torch.arange(32, dtype=torch.float32)[None, :]
    b = torch.where(k < 8, 32 * k + n, -(32 * (k - 8) + n + 1))
    return torch.cat([b, torch.zeros(16, 32)], 0)


class Packed:
    """bf16 copies of the weights in the kernels' layouts (rebuilt whenever the fp32 masters change)."""

    def __init__(self, linears, head, in_ch: int, skip: int):
        dev = head.weight.device
        self.in_ch, self.in_pad, self.skip, self.depth = in_ch, (in_ch + 31) & ~31, skip, len(linears)
        self.out_ch = head.weight.shape[0]
        pad = self.in_pad - in_ch
        self.w, self.wt, self.b = [], [], []
        for l, lin in enumerate(linears):
            w = lin.weight.detach()
            if l == 0:
                w = torch.nn.functional.pad(w, (0, pad))
            elif l == skip + 1:
                w = torch.cat([torch.nn.functional.pad(w[:, :in_ch], (0, pad)), w[:, in_ch:]], 1)
            self.w.append(w.to(torch.bfloat16).contiguous())
            # transposed copy of the part that multiplies the hidden vector: (K_h = 256 rows of k, 256 columns of n)
            wh = lin.weight.detach()[:, in_ch:] if l == skip + 1 else lin.weight.detach()
            self.wt.append(wh.t().to(torch.bfloat16).contiguous() if l > 0 else None)
            self.b.append(lin.bias.detach().float().contiguous())
        wo = torch.zeros(32, 256, device=dev)
        wo[: self.out_ch] = head.weight.detach()
        self.w_out = wo.to(torch.bfloat16).contiguous()
        self.w_out_t = head.weight.detach().t().contiguous()  # (256, out_ch) fp32, for the head's data gradient
        self.b_out = head.bias.detach().float().contiguous()
        self._wp = (C.c_void_p * self.depth)(*[t.data_ptr() for t in self.w])
        self._bp = (C.c_void_p * self.depth)(*[t.data_ptr() for t in self.b])


def forward(p: Packed, x_emb: torch.Tensor, want_acts: bool):
    N = x_emb.shape[0]
    x_emb = L.require_cuda_f32("x_emb", x_emb, (N, p.in_ch))
    out = torch.empty(N, p.out_ch, device=x_emb.device)
    acts = torch.empty(p.depth, N, 256, dtype=torch.bfloat16, device=x_emb.device) if want_acts else None
    L.check(L.lib().riggs_mlp_forward(N, p.in_ch, p.out_ch, p.depth, p.skip, p._wp, p._bp, p.w_out.data_ptr(),
                                      p.b_out.data_ptr(), x_emb.data_ptr(), L.ptr(acts), out.data_ptr(), L.stream_ptr()),
            "riggs_mlp_forward")
    return out, acts
