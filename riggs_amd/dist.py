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


# ---- zero-copy gradient bucket -------------------------------------------------------------------------
# parameter data_ptr -> (flat buffer, offset, shape): the HIP backward functions (rasterizer, PoseMLP, deformation) write
# a parameter's gradient straight into its slice of ONE flat buffer, so the all-reduce runs in place — no pack (71 MB
# torch.cat at 300k Gaussians), no divide pass, no per-step Python loop re-pointing p.grad.
_SLICES = {}  # parameter data_ptr -> (flat buffer, offset, shape, weakref to the parameter)


def _entry(param: torch.Tensor):
    """The live registry entry of ``param`` or None.  An entry dies with its parameter: densification / pruning replace
    the parameter tensors, and a NEW tensor that the allocator happens to place at the old address must not be routed
    into the old bucket."""
    ent = _SLICES.get(param.data_ptr())
    if ent is None:
        return None
    owner = ent[3]()
    if owner is None:
        del _SLICES[param.data_ptr()]
        return None
    return ent


def grad_out(param: torch.Tensor, shape=None) -> torch.Tensor:
    """The tensor a backward should write ``dL/dparam`` into: the parameter's slice of a registered
    ``FlatGradAllReduce`` bucket, else a fresh buffer.  A slice is handed out only while the parameter's ``.grad``
    does not already alias it: a second backward through the same parameter in one step (gradient accumulation, two
    renders per iteration, ``zero_grad(set_to_none=False)``) gets a fresh buffer, which autograd then ADDS to the
    live ``.grad`` — writing into the aliased slice would overwrite the first gradient and double the second."""
    shape = tuple(param.shape if shape is None else shape)
    ent = _entry(param)
    if ent is not None:
        flat, off, shp, ref = ent
        if shp == shape and flat.device == param.device:
            n = 1
            for d in shape:
                n *= d
            view = flat[off:off + n].view(shape)
            live = ref().grad
            if live is None or live.data_ptr() != view.data_ptr():
                return view
    return torch.empty(shape, dtype=torch.float32, device=param.device)


def grad_out_flat(params) -> torch.Tensor:
    """One contiguous buffer for the gradients of ``params`` in order (the PoseMLP backward writes all of its
    parameter gradients as one flat array): the bucket's own range when these parameters are registered back to
    back, else a fresh buffer."""
    total = sum(p.numel() for p in params)
    ents = [_entry(p) for p in params]
    if total and all(e is not None for e in ents):
        flat, off0 = ents[0][0], ents[0][1]
        o = off0
        ok = True
        for p, (f, off, shp, ref) in zip(params, ents):
            live = ref().grad
            ok = ok and f is flat and off == o and shp == tuple(p.shape)
            ok = ok and (live is None or live.data_ptr() != flat[off:off + 1].data_ptr())  # (see grad_out)
            o += p.numel()
        if ok and flat.device == params[0].device:
            return flat[off0:off0 + total]
    return torch.empty(total, dtype=torch.float32, device=params[0].device)


class FlatGradAllReduce:
    """ONE flat fp32 buffer holding ``dL/dp`` of every parameter, all-reduced (averaged) in place once per step.

    ``register=True`` (default on GPU) publishes the slices to the HIP backward functions (``grad_out``), which then
    write gradients directly into the bucket: ``p.grad`` IS a view of the flat buffer and ``__call__`` is just the
    collective.  Gradients that arrive in other tensors (CPU oracle pipeline, torch ops) are copied in first."""

    def __init__(self, params: Iterable[torch.Tensor], average: bool = True, register: bool = None):
        self.params: List[torch.Tensor] = list(params)
        self.average = average
        # every slice starts on a 16-byte boundary (the fused optimizer step reads gradients as float4)
        self.offsets = []
        o = 0
        for p in self.params:
            self.offsets.append(o)
            o += (p.numel() + 3) & ~3
        self.numel = o
        p0 = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]
        self.registered = p0.is_cuda if register is None else register
        if self.registered:
            self.register()

    def register(self):
        import weakref
        for p, o in zip(self.params, self.offsets):
            _SLICES[p.data_ptr()] = (self.flat, o, tuple(p.shape), weakref.ref(p))
        self.registered = True

    def __del__(self):
        try:
            self.unregister()
        except Exception:
            pass

    def unregister(self):
        for p in self.params:
            ent = _SLICES.get(p.data_ptr())
            if ent is not None and ent[0] is self.flat:
                del _SLICES[p.data_ptr()]
        self.registered = False

    def __call__(self, sources=None):
        """``sources``: gradient tensors to reduce (default: each parameter's ``.grad``).  Tensors that already are
        the bucket's slices cost nothing; anything else is copied into place.  Afterwards ``p.grad`` views the
        averaged flat buffer."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        if sources is None or sources is not getattr(self, "_in_place", None):
            src = [p.grad for p in self.params] if sources is None else sources
            in_place = True
            for g, v, p in zip(src, self.views, self.params):
                if g is None:
                    v.zero_()
                    in_place = False
                elif g.data_ptr() != v.data_ptr():
                    v.copy_(g.reshape(v.shape))
                    in_place = False
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    p.grad = v
            # a static list of gradient tensors (a captured hipGraph's) that already ARE the slices: skip this loop next time
            self._in_place = sources if (in_place and sources is not None) else None
        if world > 1:
            if self.average and self.flat.is_cuda and dist.get_backend() == "nccl":
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)  # RCCL averages in the collective: no divide pass
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                if self.average:
                    self.flat.div_(world)
        return self.flat


def frame_for_rank(frames: list, step: int, rank: int, world: int):
    """Frame k of a step's batch of `world` frames goes to rank k (one frame per GPU per step)."""
    return frames[(step * world + rank) % len(frames)]
