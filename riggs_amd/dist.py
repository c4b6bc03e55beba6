"""Frame-sharded data parallelism: every rank holds a full replica of the
Gaussian and skeleton parameters and renders its own frame; the ONLY exchange is one all-reduce
(sum, then /world) of a flat fp32 gradient buffer over RCCL/xGMI (``backend="nccl"`` on ROCm).
The reference is single-GPU (no torch.distributed call site exists in it); 8 frames per step is
batch-8 SGD, so gradients are averaged and per-frame outputs stay identical to single-GPU.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Packs ``p.grad`` of all parameters into one contiguous buffer, all-reduces it once, and hands
    back views (so the optimizer sees averaged gradients without an unpack copy)."""

    def __init__(self, params: Iterable[torch.Tensor], average: bool = True):
        self.params: List[torch.Tensor] = list(params)
        self.average = average
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def __call__(self, sources=None):
        """``sources``: gradient tensors to reduce (default: each parameter's ``.grad``) — e.g. the static
        gradient buffers owned by a captured hipGraph."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.empty(self.numel, dtype=torch.float32, device=p0.device)
        src = [p.grad for p in self.params] if sources is None else sources
        grads = [(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(src, self.params)]
        torch.cat(grads, out=self.flat)
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if self.average:
                self.flat.div_(world)
        o = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[o:o + n].view_as(p)
            o += n
        return self.flat


def frame_for_rank(frames: list, step: int, rank: int, world: int):
    """Frame k of a step's batch of `world` frames goes to rank k (one frame per GPU per step)."""
    return frames[(step * world + rank) % len(frames)]
