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


def in_bucket(param: torch.Tensor, tensor: torch.Tensor) -> bool:
    """Is ``tensor`` the slice of a registered bucket that belongs to ``param`` (i.e. memory that lives outside any
    hipGraph pool and that nothing else is ever placed in)?"""
    ent = _entry(param)
    return ent is not None and tensor is not None and tensor.data_ptr() == ent[0].data_ptr() + 4 * ent[1]


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


def _spread_pose_mlp_chain(on_gpu: bool):
    """Exchanges overlap RCCL kernels with the deformation backward: the one-launch PoseMLP kernels then must not need every
    compute unit of one XCD (include/riggs_hip.h: riggs_pose_mlp_set_placement)."""
    if on_gpu and dist.is_initialized() and dist.get_world_size() > 1:
        from . import _lib as L
        L.lib().riggs_pose_mlp_set_placement(0)


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
        # four spare floats behind the gradients travel with every collective over ``flat``: tail[0] is the step's VALIDITY
        # slot (``publish_validity``: 1 when this rank's frame went wrong; after the sum it is non-zero on every rank)
        self.flat = torch.zeros(self.numel + 4, dtype=torch.float32, device=p0.device)
        self.tail = self.flat[self.numel:]
        self.frame_gate = None  # riggs_amd._lib.FrameGate over THIS rank's frame (GraphedFrame.gate_sources())
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]
        self.registered = p0.is_cuda if register is None else register
        if self.registered:
            self.register()

    def verify_aliases(self, params=None):
        """The exchanges that reduce ``flat`` directly (OverlappedExchange, SparseRowExchange over bucket views) are only right
        when every parameter's ``.grad`` IS its slice of the bucket: a gradient that landed in a fresh buffer (grad_out falls
        back to one when the parameter is not registered, grad_out_flat when the run of parameters is not contiguous in the
        bucket) would silently stay unreduced.  Raises naming the first stray gradient; parameters without a gradient yet
        are skipped."""
        for p, v in zip(self.params, self.views):
            if params is not None and not any(p is q for q in params):
                continue
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                raise RuntimeError("the gradient of a %s parameter does not alias its slice of the flat bucket: the direct "
                                   "exchange would leave it unreduced (reduce with FlatGradAllReduce.__call__, which copies "
                                   "stray gradients in, or register the bucket before the backward)" % (tuple(p.shape),))

    def publish_validity(self):
        """In front of the collective that carries ``tail``: this rank's frame status into the validity slot (one tiny launch;
        nothing without a ``frame_gate``).  The NaN of a poisoned frame reaches every rank's gradients through the sum — and so
        does this flag, which every rank's optimizer takes as a gate word (``validity_source``): all replicas skip that step."""
        if self.frame_gate is not None and self.tail is not None and self.flat.is_cuda:
            from . import _lib as L
            import ctypes as C
            L.check(L.lib().riggs_gate_flag(C.byref(self.frame_gate.struct()), self.tail.data_ptr(), L.stream_ptr()), "riggs_gate_flag")

    def validity_source(self):
        """(tensor, word index, mask) of the validity slot, for the ``FrameGate`` of the optimizers that step on this bucket; None
        when the bucket has no slot (``ShardedAdam`` moves the gradients into its own padded buffer without one): a FrameGate
        skips sources that return None."""
        if self.tail is None:
            return None
        return (self.flat, self.numel, 0x7FFFFFFF)

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
        self.publish_validity()
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


# ---- overlapped exchange ------------------------------------------------------------------------------------------
class OverlappedExchange:
    """The data-parallel exchange of a 0.4 ms step, as two phases of the flat bucket issued on a communication stream:

    * phase 1 — ``bucket.flat[:split]``: the gradients that are FINAL when the rasterizer's per-Gaussian backward has run
      (SH, opacity, scale: 86 % of the bytes).  Its all-reduce is issued right there and runs on the xGMI links while the
      compute stream still does the deformation backward (skinning, FK, PoseMLP: ~60 us of the step);
    * phase 2 — the rest: ``_xyz`` / ``_rotation`` gradients — which the deformation backward still READS as dL/dd_xyz and
      dL/dd_rotation (they alias the parameter gradients), so they must not be averaged in place under it — and the
      skeleton's own parameters, issued after the deformation backward.

    ``chunk_bytes`` splits a phase into several collectives (RCCL pipelines inside one call; smaller calls let the first
    bytes leave earlier).  The reference has no distributed path (utils/general_utils.py:207 pins cuda:0): this is new."""

    def __init__(self, bucket: FlatGradAllReduce, split: int, chunk_bytes: int = 0):
        self.bucket, self.split = bucket, int(split)
        self.chunk = int(chunk_bytes) // 4
        self.cuda = bucket.flat.is_cuda
        _spread_pose_mlp_chain(self.cuda)
        self.comm = torch.cuda.Stream(device=bucket.flat.device) if self.cuda else None
        self.pending = []

    def _reduce(self, t):
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return
        pieces = [t] if self.chunk <= 0 else [t[o:o + self.chunk] for o in range(0, t.numel(), self.chunk)]
        for piece in pieces:
            if self.bucket.average and self.cuda and dist.get_backend() == "nccl":
                self.pending.append((dist.all_reduce(piece, op=dist.ReduceOp.AVG, async_op=True), None))
            else:
                self.pending.append((dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True), piece if self.bucket.average else None))

    def launch(self, phase: int):
        """Issue phase 1 (``flat[:split]``) or phase 2 (``flat[split:]``) behind everything the compute stream has queued."""
        t = self.bucket.flat[:self.split] if phase == 1 else self.bucket.flat[self.split:]
        if phase == 1 and not (self.cuda and torch.cuda.is_current_stream_capturing()):
            self.bucket.verify_aliases()
        if phase != 1:
            self.bucket.publish_validity()  # (phase 2 carries the bucket's tail: the frame is complete by now)
        if self.cuda:
            self.comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm):
                self._reduce(t)
        else:
            self._reduce(t)

    def wait(self):
        """Make the compute stream (and, on CPU, the caller) wait for every collective issued so far."""
        world = dist.get_world_size() if dist.is_initialized() else 1
        ctx = torch.cuda.stream(self.comm) if self.cuda else None
        if ctx is not None:
            ctx.__enter__()
        try:
            for work, needs_div in self.pending:
                work.wait()
                if needs_div is not None:
                    needs_div.div_(world)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
        self.pending = []
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.comm)


def exchange_order(gm, sw):
    """Parameters in the order the overlapped exchange wants them in ONE flat bucket, and the number of leading tensors whose
    gradients are final after the rasterizer's backward (phase 1): SH, opacity and scale first; then ``_xyz`` / ``_rotation``
    (read by the deformation backward) and the skeleton's parameters."""
    first = [gm._features_dc, gm._features_rest, gm._opacity, gm._scaling]
    rest = [gm._xyz, gm._rotation, sw._node_radius] + list(sw.pose_net.parameters())
    return first + rest, len(first)


def row_exchange_order(gm, sw):
    """Parameters in the order the gradient-row exchange wants them in ONE flat bucket — the six per-Gaussian tensors
    (their gradients come from the rasterizer's backward alone: ``SparseRowExchange.rows``), then the skeleton's (dense,
    0.6 M floats: ``rest``) — and the number of per-Gaussian tensors."""
    rows = [gm._features_dc, gm._features_rest, gm._opacity, gm._scaling, gm._xyz, gm._rotation]
    return rows + [sw._node_radius] + list(sw.pose_net.parameters()), len(rows)


# ---- sharded optimizer step (ZeRO-1 over the frame-parallel replicas) -----------------------------------------------
class ShardedAdam:
    """reduce-scatter -> Adam on the rank's 1/W slice of the flat parameter space -> all-gather: the same
    bytes on the links as the all-reduce (its two halves), the optimizer's HBM traffic (28 B per element: the largest consumer
    after the frame itself) divided by the world size, and the Adam moments held once per node instead of once per GPU.

    Parameters become views of ONE flat buffer (so the all-gather lands in place) and their gradients live in the matching
    ``FlatGradAllReduce`` bucket.  ``groups`` = [{"params": [...], "lr": float}, ...] as for torch.optim.Adam (plain Adam,
    eps inside the square root's sum as torch: the update of riggs_adam_step / torch.optim.Adam).  After densification /
    pruning (new parameter tensors) build a new instance: the moments are per flat element."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-15, average=True):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.b1, self.b2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        # the groups are kept and their "lr" is read at EVERY step, like torch.optim.Adam's param_groups: the reference
        # trainer rewrites the xyz learning rate every iteration (scene/gaussian_model.py: update_learning_rate)
        self.param_groups = [dict(g, params=list(g["params"])) for g in groups]
        self.params = [p for g in self.param_groups for p in g["params"]]
        self._group_of = [gi for gi, g in enumerate(self.param_groups) for _ in g["params"]]
        self.bucket = FlatGradAllReduce(self.params, average=average, register=False)
        n = self.bucket.numel
        self.shard = (n + 4 * self.world - 1) // (4 * self.world) * 4     # 16-byte aligned, equal on every rank
        dev = self.params[0].device
        self.flat_p = torch.zeros(self.shard * self.world, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(self.shard * self.world, dtype=torch.float32, device=dev)
        self.flat_g[:n] = self.bucket.flat[:n]
        # the bucket's gradient buffer and the parameters move into the padded flat buffers (views keep their offsets)
        self.bucket.flat, self.bucket.tail = self.flat_g[:n], None
        self.bucket.views = [self.bucket.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.bucket.offsets)]
        with torch.no_grad():
            for p, o in zip(self.params, self.bucket.offsets):
                self.flat_p[o:o + p.numel()] = p.detach().reshape(-1)
                p.data = self.flat_p[o:o + p.numel()].view_as(p)
        if dev.type == "cuda":
            self.bucket.register()  # (keyed by the parameters' NEW data pointers: the HIP backwards write into the bucket)
        lo = self.rank * self.shard
        self.lo, self.hi = lo, lo + self.shard
        self.m = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.steps = 0
        # (tensor, slice of it that falls into this rank's shard): the segments one optimizer launch covers
        self.segments = []
        for p, o, lr_i in zip(self.params, self.bucket.offsets, range(len(self.params))):
            a, b = max(o, self.lo), min(o + p.numel(), self.hi)
            if a < b:
                self.segments.append((a, b, lr_i))

    @property
    def lrs(self):
        return [float(self.param_groups[gi]["lr"]) for gi in self._group_of]

    def set_lr(self, group_name_or_index, lr: float):
        """The schedule hook (``update_learning_rate``): by group index or by the group's "name"."""
        for gi, g in enumerate(self.param_groups):
            if gi == group_name_or_index or g.get("name") == group_name_or_index:
                g["lr"] = float(lr)
                return
        raise KeyError(group_name_or_index)

    def step(self):
        """Gradients (already in the bucket) -> averaged shard -> Adam on the shard -> every replica's parameters."""
        g_shard = self.flat_g[self.lo:self.hi]
        if self.world > 1:
            if self.flat_g.is_cuda and dist.get_backend() == "nccl":
                out = torch.empty_like(g_shard)
                dist.reduce_scatter_tensor(out, self.flat_g, op=dist.ReduceOp.AVG if self.bucket.average else dist.ReduceOp.SUM)
                g_shard = out
            else:  # gloo has no reduce-scatter: all-reduce and look at the own slice (CPU tests)
                dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM)
                if self.bucket.average:
                    self.flat_g.div_(self.world)
                g_shard = self.flat_g[self.lo:self.hi]
        self.steps += 1
        p_shard = self.flat_p[self.lo:self.hi]
        if p_shard.is_cuda:
            self._hip_step(p_shard, g_shard)
        else:
            bc1, bc2 = 1.0 - self.b1 ** self.steps, 1.0 - self.b2 ** self.steps
            with torch.no_grad():
                self.m.mul_(self.b1).add_(g_shard, alpha=1.0 - self.b1)
                self.v.mul_(self.b2).addcmul_(g_shard, g_shard, value=1.0 - self.b2)
                for a, b, i in self.segments:
                    sl = slice(a - self.lo, b - self.lo)
                    denom = (self.v[sl].sqrt() / (bc2 ** 0.5)).add_(self.eps)
                    p_shard[sl].addcdiv_(self.m[sl], denom, value=-float(self.lrs[i]) / bc1)
        if self.world > 1:
            if self.flat_p.is_cuda and dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(self.flat_p, p_shard)
            else:
                self._gather_gloo(p_shard)
        for p in self.params:
            torch.autograd.graph.increment_version(p)

    def _gather_gloo(self, p_shard):
        parts = [torch.empty_like(p_shard) for _ in range(self.world)]
        dist.all_gather(parts, p_shard.clone())
        self.flat_p.copy_(torch.cat(parts))

    def _hip_step(self, p_shard, g_shard):
        import ctypes as C
        from . import _lib as L
        lib = L.lib()
        segs = self.segments
        for c0 in range(0, len(segs), 32):
            chunk = segs[c0:c0 + 32]
            n = len(chunk)
            off = lambda t, a: t.data_ptr() + 4 * (a - self.lo)  # noqa: E731
            P = (C.c_void_p * n)(*[off(p_shard, a) for a, _, _ in chunk])
            G = (C.c_void_p * n)(*[off(g_shard, a) for a, _, _ in chunk])
            M = (C.c_void_p * n)(*[off(self.m, a) for a, _, _ in chunk])
            V = (C.c_void_p * n)(*[off(self.v, a) for a, _, _ in chunk])
            numel = (C.c_int64 * n)(*[b - a for a, b, _ in chunk])
            lrs = self.lrs
            lr = (C.c_double * n)(*[lrs[i] for _, _, i in chunk])
            steps = (C.c_int64 * n)(*[self.steps] * n)
            L.check(lib.riggs_adam_step(n, P, G, M, V, numel, lr, steps, self.b1, self.b2, self.eps, L.stream_ptr()),
                    "riggs_adam_step")


# ---- compacted exchange of the rows that have a gradient --------------------------------------------------------
def sparse_rows_all_reduce(grads, capacity: int, average: bool = True):
    """All-reduce of per-Gaussian gradient tensors ``grads`` = [(N, ...), ...] that exchanges only the Gaussians with a non-zero
    gradient on some rank: in the bench scene 93 % of the rows are zero on every rank (SURVEY.md §8-d scene: deep, mostly
    occluded), and different views touch different Gaussians.  Every rank compacts its non-zero rows (fixed ``capacity``: no
    host synchronisation), all-gathers (index, row) pairs and adds them up locally: (W - 1) x capacity x (row bytes + 4) received
    per rank instead of 2 (W - 1) / W x N x row bytes for the ring all-reduce — less traffic while capacity < ~2 N / W.
    Returns the number of rows the fullest rank needed (a device scalar: when it exceeds ``capacity`` the gradients are left
    UNTOUCHED — local — on every rank and the caller must redo the exchange densely; checked like the rasterizer's arena
    overflow, after the fact)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    N = grads[0].shape[0]
    flat = [g.reshape(N, -1) for g in grads]
    nz = torch.zeros(N, dtype=torch.bool, device=flat[0].device)
    for f in flat:
        nz |= (f != 0).any(1)
    need = nz.sum()
    idx = torch.nonzero_static(nz, size=capacity, fill_value=N).reshape(-1)          # (capacity,), N = "no row"
    safe = idx.clamp(max=N - 1)
    valid = (idx < N).to(flat[0].dtype)[:, None]
    payload = torch.cat([f.index_select(0, safe) for f in flat], dim=1) * valid       # (capacity, sum of widths)
    if world > 1:
        all_idx = torch.empty(world * capacity, dtype=idx.dtype, device=idx.device)
        all_pay = torch.empty(world * capacity, payload.shape[1], dtype=payload.dtype, device=payload.device)
        if idx.is_cuda and dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(all_idx, idx)
            dist.all_gather_into_tensor(all_pay, payload)
        else:
            li, lp = [torch.empty_like(idx) for _ in range(world)], [torch.empty_like(payload) for _ in range(world)]
            dist.all_gather(li, idx)
            dist.all_gather(lp, payload)
            all_idx, all_pay = torch.cat(li), torch.cat(lp)
        dist.all_reduce(need, op=dist.ReduceOp.MAX)
    else:
        all_idx, all_pay = idx, payload
    acc = torch.zeros(N + 1, all_pay.shape[1], dtype=all_pay.dtype, device=all_pay.device)
    acc.index_add_(0, all_idx, all_pay)
    if average:
        acc /= world
    # on overflow (some rank needed more than `capacity` rows: `need` is the maximum over the ranks, the same on all of them) the
    # compaction dropped rows, so NOTHING is written back — every rank keeps its local gradients and can still redo the step
    # densely; decided on the device, no host synchronisation
    ok = need <= capacity
    o = 0
    for g, f in zip(grads, flat):
        w = f.shape[1]
        g.copy_(torch.where(ok, acc[:N, o:o + w].reshape(g.shape), g))
        o += w
    return need


# ---- gradient-row exchange (csrc/exchange.hip) --------------------------------------------------------------------------
def segment_words(N: int, row_floats: int, capacity: int) -> int:
    """32-bit words of one rank's segment (include/riggs_hip.h: riggs_grad_rows_segment_bytes / 4)."""
    rows_off = (4 + (N + 255) // 256 + 1 + 3) // 4 * 4
    return ((rows_off + capacity * row_floats) * 4 + 255) // 256 * 256 // 4


def _hip_pack(ex):
    import ctypes as C
    from . import _lib as L
    from .rasterizer import last_backward_workspace
    ws, n = (ex.workspace, ex.N) if ex.workspace is not None else last_backward_workspace()
    if n != ex.N or ws.device != ex.segment.device:
        raise RuntimeError("the last rasterizer backward was not over these %d Gaussians" % ex.N)
    scale = C.c_float(1.0 / ex.world if ex.average else 1.0)
    if ex.gate is not None:  # an invalid frame marks its segment instead of packing: every rank then skips this step
        L.check(L.lib().riggs_grad_rows_pack_gated(ex.N, ws.data_ptr(), len(ex.rows), ex._ptrs, ex._widths, scale, ex.capacity,
                                                   ex.segment.data_ptr(), C.byref(ex.gate.struct()), L.stream_ptr()),
                "riggs_grad_rows_pack_gated")
        return
    L.check(L.lib().riggs_grad_rows_pack(ex.N, ws.data_ptr(), len(ex.rows), ex._ptrs, ex._widths, scale,
                                         ex.capacity, ex.segment.data_ptr(), L.stream_ptr()), "riggs_grad_rows_pack")


def _hip_unpack(ex):
    from . import _lib as L
    ws = ex.workspace.data_ptr() if (ex.workspace is not None and ex.record_rows) else None
    L.check(L.lib().riggs_grad_rows_unpack(ex.N, ex.world, ex.capacity, ex.gathered.data_ptr(), len(ex.rows), ex._ptrs, ex._widths,
                                           ex.status.data_ptr(), ws, L.stream_ptr()), "riggs_grad_rows_unpack")


class SparseRowExchange:
    """The data-parallel exchange of a frame whose per-Gaussian gradients are sparse BY ROW (93 % of the rows are exactly
    zero in the SURVEY.md §8-d scene): pack the touched rows (csrc/exchange.hip, riggs_grad_rows_pack — the list comes from
    the rasterizer's backward, nothing is scanned), all-gather the packed segments, combine them in rank order on every
    rank (riggs_grad_rows_unpack: no atomics, bit-identical replicas).  Per link and direction a rank sends capacity x 240
    bytes once to every peer — CONSTANT in the world size on the xGMI mesh — where the ring all-reduce of the dense bucket
    moves 2 N x 236 / W bytes: the rows win while capacity < 2 N / W (DESIGN.md §5 has the budget).

    ``rows`` — the gradient tensors (N, ...) of the per-Gaussian parameters, fixed storage (bucket views / the captured
    graph's gradient buffers); every one of them must get its gradient ONLY from the rasterizer's backward (a regulariser
    that touches other rows needs the dense path).  ``rest`` — a flat tensor with the remaining (dense, small) gradients,
    all-reduced as it is.  ``pack()`` right after the rasterizer's backward on the compute stream, ``launch()`` puts the
    all-gather on the communication stream, ``launch_rest()`` the small all-reduce (after the deformation backward),
    ``wait()`` joins and unpacks.  ``check()`` (a device->host read of a STICKY status) returns False when a segment
    overflowed ``capacity`` in any step since the previous check: that step's unpack was skipped on EVERY rank (gradients
    still local).  Checked before the optimizer steps, ``dense_fallback()`` repairs the step; checked only every k steps,
    the replicas may already have stepped on un-averaged gradients — ``resync()`` them; then build a larger exchange
    (``resize()``, before any capture).  The reference has no distributed path."""

    @staticmethod
    def rows_on_gpu(rows):
        return len(rows) > 0 and rows[0].is_cuda

    def __init__(self, rows, rest=None, capacity=None, average=True, pack=None, unpack=None, world=None,
                 force_collectives=False, validity=None):
        """``force_collectives``: issue the collectives even on a communicator of ONE rank (where they are the identity) —
        what ``bench.py``'s single-GPU exchange-path measurement uses to run the real RCCL enqueue path on one device."""
        self.world = int(world) if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.force = bool(force_collectives) and dist.is_initialized() and world is None
        # the FlatGradAllReduce whose tail ``rest`` ends with (rest = bucket.flat[offset:]): its validity slot is published in
        # front of the dense all-reduce of ``rest`` — a PoseMLP hand-off lost in the deformation BACKWARD happens after the
        # pack, poisons only the skeleton's gradients, and reaches the other ranks through that sum
        self.validity = validity
        _spread_pose_mlp_chain(self.rows_on_gpu(rows))
        self.rows = [g for g in rows]
        self.N = int(self.rows[0].shape[0])
        if any(g.shape[0] != self.N or not g.is_contiguous() or g.dtype != torch.float32 for g in self.rows):
            raise ValueError("rows: contiguous float32 tensors with one row per Gaussian")
        if not 1 <= len(self.rows) <= 8:
            raise ValueError("1..8 row tensors")
        self.widths = [g.numel() // self.N for g in self.rows]
        self.row_floats = (1 + sum(self.widths) + 3) // 4 * 4
        self.rest, self.average = rest, bool(average)
        self.cuda = self.rows[0].is_cuda
        if not self.cuda and (pack is None or unpack is None):
            raise RuntimeError("SparseRowExchange packs / unpacks with HIP kernels: CUDA tensors required")
        self._pack, self._unpack = pack or _hip_pack, unpack or _hip_unpack
        import ctypes as C
        self._ptrs = (C.c_void_p * len(self.rows))(*[g.data_ptr() for g in self.rows])
        self._widths = (C.c_int32 * len(self.rows))(*self.widths)
        dev = self.rows[0].device
        self.comm = torch.cuda.Stream(device=dev) if self.cuda else None
        self.status = torch.zeros(8, dtype=torch.int32, device=dev)  # [0..3] sticky: see check(); [4] this step's flag (the gate's word)
        self.pending = []
        self.need, self.calls, self.overflow_call = 0, 0, 0
        self.invalid_frame = False  # as of the last check(): some rank's frame was invalid in a step since the previous one
        # riggs_amd._lib.FrameGate over THIS rank's frame (GraphedFrame.gate_sources()): pack() then marks the segment
        # "frame invalid" instead of packing when the frame went wrong (GraphedFrame.capture_exchange sets it)
        self.gate = None
        self._pack_captured = False
        # the backward workspace to pack from: None = the one the most recent rasterizer backward used; a caller that runs
        # other backward passes in between (another stream, an eager profiling step) pins it (GraphedFrame.backward_workspace)
        self.workspace = None
        # with a pinned workspace: record the rows the unpack writes in it, so that a captured frame whose backward skips
        # the zero fill of untouched rows (GraphedFrame(sparse_grad_rows=True)) zeroes the rows other ranks touched
        self.record_rows = False
        # the small dense all-reduce gets its OWN communicator: collectives of one process group run one after the other
        # on that group's internal stream, and this one must not queue behind the all-gather of the rows
        self.rest_group = None
        if rest is not None and (self.world > 1 or self.force) and dist.is_initialized() and world is None:
            self.rest_group = dist.new_group()
        self.resize(int(capacity) if capacity is not None else max(1024, self.N // 8))

    def resize(self, capacity: int):
        if getattr(self, "_pack_captured", False):
            # a captured pack has the old segment's address and capacity baked in: it would write into freed memory while the
            # all-gather sends the new, empty segment (gradients silently stay local)
            raise RuntimeError("SparseRowExchange.resize() after pack() was captured in a hipGraph: build a new exchange and "
                               "capture the frame again")
        self.capacity = int(min(max(capacity, 1), self.N))
        words = segment_words(self.N, self.row_floats, self.capacity)
        dev = self.rows[0].device
        self.segment = torch.zeros(words, dtype=torch.int32, device=dev)
        self.gathered = torch.zeros(self.world * words, dtype=torch.int32, device=dev)

    @property
    def wins(self) -> bool:
        """Rows beat the dense ring all-reduce on the links while capacity < 2 N / W (always at W = 2)."""
        return self.capacity * self.world < 2 * self.N

    def pack(self):
        if self.cuda and torch.cuda.is_current_stream_capturing():
            self._pack_captured = True
        self._pack(self)

    def _capturing(self):
        return self.cuda and torch.cuda.is_current_stream_capturing()

    def _on_comm(self, fn):
        # Inside a hipGraph capture the collectives are issued from the capturing stream itself: the process group forks to
        # its internal stream and joins at ``wait()``, which the capture records as a branch of the graph — the overlap with
        # the deformation backward is the same.  (A fork to a user side stream around an RCCL call crashes
        # hipStreamEndCapture on ROCm 7.2 / torch 2.10: tools/scratch/rccl_capture_probe.py.)
        if self.cuda and not self._capturing():
            self.comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm):
                fn()
        else:
            fn()

    def launch(self):
        def go():
            if self.world == 1 and not self.force:
                self.gathered.copy_(self.segment)
            elif self.cuda and dist.get_backend() == "nccl":
                self.pending.append((dist.all_gather_into_tensor(self.gathered, self.segment, async_op=True), None))
            elif self.cuda:  # gloo (the CPU-backend control-flow tests): its all_gather has no device implementation
                host = self.segment.cpu()
                parts = [torch.empty_like(host) for _ in range(self.world)]
                dist.all_gather(parts, host)
                self.gathered.copy_(torch.cat(parts))
            else:
                parts = list(self.gathered.view(self.world, -1).unbind(0))
                self.pending.append((dist.all_gather(parts, self.segment, async_op=True), None))
        self._on_comm(go)

    def _reduce_dense(self, t, group=None):
        if self.world == 1 and not self.force:
            return
        if self.average and self.cuda and dist.get_backend() == "nccl":
            self.pending.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group, async_op=True), None))
        else:
            self.pending.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True), t if self.average else None))

    def launch_rest(self):
        if self.rest is not None:
            if self.validity is not None:
                self.validity.publish_validity()
            self._on_comm(lambda: self._reduce_dense(self.rest, self.rest_group))

    def _join(self):
        def go():
            for work, needs_div in self.pending:
                work.wait()
                if needs_div is not None:
                    needs_div.div_(self.world)
            self.pending = []
        if self.cuda and not self._capturing():
            with torch.cuda.stream(self.comm):
                go()
            torch.cuda.current_stream().wait_stream(self.comm)
        else:
            go()

    def wait(self):
        self._join()
        self._unpack(self)

    def status_source(self):
        """For the ``FrameGate`` of the optimizers that consume the exchanged gradients: THIS step's status word (rewritten by
        every unpack) — raised on EVERY rank in the same step when a segment overflowed or some rank's frame was invalid
        (nothing was unpacked).  Gated on it, all replicas skip exactly that step together — they stay bit-identical without a
        ``resync()`` — and go on with the next good one; the sticky words ``check()`` reads keep the report for the host."""
        return (self.status, 4, 0xFFFFFFFF)

    def check(self) -> bool:
        """Reads and clears the sticky status (a device->host read).  False when ANY unpack since the previous check
        overflowed ``capacity`` — on such a step every rank skipped the unpack, the gradients stayed local — or when some
        rank's frame was invalid (``invalid_frame``: its pack was gated, every rank skipped).
        ``overflow_call`` then holds which of the ``calls`` unpacks since the previous check failed first: if it is the one
        that has just run and no optimizer has stepped, ``dense_fallback()`` repairs the step; if optimizers have stepped on
        it (polling every k steps), the replicas have diverged: ``resync(parameters, optimizer)`` broadcasts rank 0's."""
        need, bad, calls, first = (int(v) for v in self.status.tolist()[:4])
        self.status.zero_()
        self.need, self.calls, self.overflow_call = need, calls, first if bad else 0
        self.invalid_frame = bool(bad & 2)
        return not bad

    def resync(self, tensors):
        """Replicas that stepped on un-averaged gradients (an overflow found late): every tensor in ``tensors`` — parameters,
        optimizer moments, step counters — becomes rank 0's, then ``resize()`` to what the steps needed."""
        if self.world > 1 and dist.is_initialized():
            for t in tensors:
                if t.is_cuda and dist.get_backend() != "nccl":
                    h = t.detach().cpu()
                    dist.broadcast(h, 0)
                    t.detach().copy_(h)
                else:
                    dist.broadcast(t.detach(), 0)

    def dense_fallback(self):
        """The step whose ``check()`` failed: its unpack was skipped everywhere, so the rows are averaged densely."""
        for g in self.rows:
            self._reduce_dense(g)
        self._join()
        if self.record_rows and self.workspace is not None:  # every row may now hold something: the next sparse backward rewrites all
            from .rasterizer import mark_all_rows
            mark_all_rows(self.workspace, self.N)
