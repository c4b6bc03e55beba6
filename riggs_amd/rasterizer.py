"""Host side of the HIP rasterizer: the `diff_gaussian_rasterization` Python surface.

Mirrors what RigGS imports at /root/reference/gaussian_renderer/__init__.py:14 and calls
at :57-72 (GaussianRasterizationSettings, by keyword) and :133-141
(GaussianRasterizer.forward, by keyword) — same names, argument meaning, 4-tuple result
``(color (3,H,W), radii (N,) int32, depth (1,H,W), alpha (1,H,W))`` and error behaviour.
All compute is in libriggs_hip.so; torch provides memory, streams and autograd plumbing.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib as L


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class RasterArena:
    """Persistent instance arena: lets consecutive frames skip the device->host read of the
    instance count R (upstream synchronises on it every call).  The arena is sized from the
    previous frame (x1.25 head-room); the kernels are overflow-safe and the (rare) overflow is
    detected when the count is finally read, then the frame is re-rendered with a larger arena.
    """

    def __init__(self, growth: float = 1.25, min_capacity: int = 1 << 16, tight_lists: bool = False):
        self.growth = growth
        # riggs_raster_cfg.tight_lists for the frames rendered through this arena: tile rectangles cut down to the tiles in which
        # the Gaussian can reach alpha >= 1/255 (its axis-aligned alpha box instead of the ceil(3 sigma) square).  The dropped
        # instances fail the alpha test at every pixel of their tile, so radii, images and gradients are the canonical ones;
        # about a fifth fewer instances to sort, stage and walk in a translucent scene.  Off by default: the canonical lists —
        # upstream's, bit for bit — are what north_star's ordering / indexing parity is stated on.  It travels with the arena
        # (the frame's persistent state), like ``sparse_grad_rows``; there is no module-wide switch.
        self.tight_lists = bool(tight_lists)
        self.capacity = 0
        self.min_capacity = min_capacity
        self.binning: Optional[torch.Tensor] = None
        self.last_R = -1
        self._dims = None     # (N, H, W, device) of the last frame
        self._ring, self._ring_at = [], -1  # recycled (event, pinned host counters) pairs of the asynchronous read-back
        self._fits = None     # (capacity, N, H, W) the current allocation is known to hold
        self._pending = None  # (event, pinned host counters, capacity used)
        self.static_counters = None
        # set by the OWNER of a captured frame (riggs_amd.graph.GraphedFrame(sparse_grad_rows=True)): backwards of frames
        # rendered through THIS arena inside a hipGraph capture may skip the zero fill of gradient rows that have no gradient
        # now and had none in the previous replay (riggs_raster_cfg.sparse_zero).  It travels with the arena — the object that
        # is the frame's persistent state — instead of being a module-wide switch toggled around the capture.
        self.sparse_grad_rows = False
        self._layout_key = None  # (capacity, N, H, W) the arena's walk history belongs to

    def _post(self, counters: torch.Tensor, cap: int):
        """Queue an asynchronous read-back of (R, overflow) behind the frame just launched."""
        self.static_counters = counters
        if torch.cuda.is_current_stream_capturing():
            return  # inside a hipGraph capture: GraphedFrame.check() reads the counters after the replay
        # (pinned buffers and events are recycled: allocating a pinned tensor and an event per frame was 15 us of an eager frame)
        ring = self._ring
        if len(ring) < 4:
            ring.append((torch.cuda.Event(), torch.empty(4, dtype=torch.int32, pin_memory=True)))
        self._ring_at = (self._ring_at + 1) % len(ring)
        ev, host = ring[self._ring_at]
        if not ev.query():  # (its copy of four frames ago is still in flight: a read-back nobody resolved — leave that pair alone)
            ev, host = torch.cuda.Event(), torch.empty(4, dtype=torch.int32, pin_memory=True)
        host.copy_(counters, non_blocking=True)
        ev.record()
        self._pending = (ev, host, cap)

    def resolve(self, block: bool = True) -> bool:
        """Consume the pending read-back of the previous frame (normally long complete: no stall).
        Raises when that frame overflowed the arena — its image and gradients are invalid."""
        if self._pending is None:
            return True
        ev, host, cap = self._pending
        if not block and not ev.query():
            return False
        ev.synchronize()
        self._pending = None
        c = host.tolist()
        R, overflow = c[0] & 0xFFFFFFFF, c[1]
        self.last_R = R
        if overflow & 2:
            raise L.RiggsHipError("the depth sort's third pass could not synchronise its workgroups on the previous frame (GPU shared "
                                  "with another long-running kernel?); that frame's outputs are invalid")
        if overflow:
            raise L.RiggsHipError("instance arena overflowed on the previous frame (R=%d > capacity=%d); that frame\'s "
                                  "outputs are invalid.  The arena is regrown on the next call." % (R, cap))
        return True

    def ensure(self, cap: int, N: int, H: int, W: int, device, minimum: Optional[int] = None):
        """The arena for ``cap`` instances (the previous frame's count with the growth head-room).  An arena that still holds
        ``minimum`` (the count with a THIRD of that head-room; default: ``cap``) is kept: a scene whose count creeps up from
        frame to frame — every training iteration — would otherwise be given a new arena, and a fresh walk history, by each
        frame, including the one a hipGraph capture records (the allocation and the history's memset then replay for ever)."""
        cap = max(int(cap), self.min_capacity)
        self._dims = (N, H, W, device)
        same = self.binning is not None and self.binning.device == device
        if same and (cap if minimum is None else min(int(minimum), cap)) <= self.capacity:
            # enough instances — but the arena also holds tables sized by the number of Gaussians and of tiles (the tile
            # sort's chunk x tile table): a scene that grew, or a larger image, needs a larger arena at the same capacity
            # (asked of the library once per (capacity, N, H, W))
            key = (self.capacity, N, H, W)
            if self._fits == key:
                return self._with_fresh_history(N, H, W)
            if L.lib().riggs_raster_binning_bytes(self.capacity, N, H, W) <= self.binning.numel():
                self._fits = key
                return self._with_fresh_history(N, H, W)
            cap = self.capacity
        self.binning = torch.empty(L.lib().riggs_raster_binning_bytes(cap, N, H, W), dtype=torch.uint8, device=device)
        self.capacity = cap
        self._layout_key = None
        self._fits = (cap, N, H, W)
        return self._with_fresh_history(N, H, W)

    def top_up(self):
        """Full head-room over the last frame's count, now (eagerly): what the owner of a hipGraph capture calls between its
        warm-up frames and the capture, so that the captured frame neither allocates nor resets the history."""
        if self._dims is not None and self.last_R >= 0:
            self.ensure(int(self.last_R * self.growth) + 1, *self._dims)

    def _with_fresh_history(self, N: int, H: int, W: int):
        """The arena's one piece of frame-to-frame state — how deep the forward walked every tile's list — sits at an offset
        that depends on (capacity, N, H, W) and is trusted when a stamp word follows it: a new allocation (torch.empty may hand
        back a block that still holds another arena's stamp) or a changed scene / image size starts without a history."""
        key = (self.capacity, N, H, W)
        if self._layout_key != key:
            L.check(L.lib().riggs_raster_binning_reset_history(self.binning.data_ptr(), self.capacity, N, H, W, L.stream_ptr()),
                    "riggs_raster_binning_reset_history")
            self._layout_key = key
        return self.binning


def _cfg(settings: GaussianRasterizationSettings, N: int, M: int, glue: bool, isotropic: bool, keep: list, tight_lists: bool = False):
    dev = settings.viewmatrix.device
    bg = settings.bg
    bg = L.require_cuda_f32("bg", (bg if bg.device == dev else bg.to(dev)).reshape(-1), (3,))
    view = L.require_cuda_f32("viewmatrix", settings.viewmatrix, (4, 4))
    proj = L.require_cuda_f32("projmatrix", settings.projmatrix, (4, 4))
    campos = L.require_cuda_f32("campos", settings.campos.reshape(-1), (3,))
    keep += [bg, view, proj, campos]
    c = L.RasterCfg()
    c.num_points = N
    c.sh_degree = int(settings.sh_degree)
    c.sh_coeffs = M
    c.image_height = int(settings.image_height)
    c.image_width = int(settings.image_width)
    c.tanfovx = float(settings.tanfovx)
    c.tanfovy = float(settings.tanfovy)
    c.scale_modifier = float(settings.scale_modifier)
    c.bg, c.viewmatrix, c.projmatrix, c.campos = bg.data_ptr(), view.data_ptr(), proj.data_ptr(), campos.data_ptr()
    c.debug = 1 if settings.debug else 0
    c.glue = 1 if glue else 0
    c.isotropic = 1 if isotropic else 0
    c.deterministic = 1 if ORDERED_BACKWARD else 0
    c.tight_lists = 1 if tight_lists else 0
    return c


ORDERED_BACKWARD = False


def set_ordered_backward(on: bool = True):
    """Reproducible mode (riggs_raster_cfg.deterministic): the compositing backward sums per-instance gradient rows per
    Gaussian in ascending tile order instead of float atomics — bitwise reproducible gradients, for tests and debugging
    (SURVEY.md §5).  It also makes the forward ignore the arena's walk history: with it the image is bitwise reproducible frame
    after frame; without it only while no tile is composited by the forward's 32-lane blocks — always in a fresh arena —,
    whose sums fold in another order (same values to ~1e-7).  Applies to rasterizations started after the call."""
    global ORDERED_BACKWARD
    ORDERED_BACKWARD = bool(on)


class _Saved:
    pass


def rasterize_forward(settings, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                      d_xyz=None, d_rotation=None, d_scaling=None, glue=False, isotropic=False,
                      arena: Optional[RasterArena] = None, shs_rest=None, tight_lists: Optional[bool] = None):
    """Runs both forward stages.  Returns (color, radii, depth, alpha, saved-state).  ``tight_lists``: riggs_raster_cfg.tight_lists
    for this frame (default: the arena's setting; False without an arena — see ``RasterArena``)."""
    if tight_lists is None:
        tight_lists = arena is not None and arena.tight_lists
    lib = L.lib()
    N = means3D.shape[0]
    dev = means3D.device
    H, W = int(settings.image_height), int(settings.image_width)
    M = 0 if shs is None else shs.shape[1] + (0 if shs_rest is None else shs_rest.shape[1])
    keep = []
    cfg = _cfg(settings, N, M, glue, isotropic, keep, tight_lists)
    geom = torch.empty(lib.riggs_raster_geom_bytes(N), dtype=torch.uint8, device=dev)
    img = torch.empty(lib.riggs_raster_image_bytes(H, W), dtype=torch.uint8, device=dev)
    radii = torch.empty(N, dtype=torch.int32, device=dev)
    counters = torch.empty(4, dtype=torch.int32, device=dev)
    color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
    depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
    st = L.stream_ptr()
    L.check(lib.riggs_raster_preprocess(C.byref(cfg), L.ptr(means3D), L.ptr(shs), L.ptr(shs_rest), L.ptr(colors_precomp),
                                        L.ptr(opacities), L.ptr(scales), L.ptr(rotations), L.ptr(cov3D_precomp),
                                        L.ptr(d_xyz), L.ptr(d_rotation), L.ptr(d_scaling), geom.data_ptr(), radii.data_ptr(),
                                        counters.data_ptr(), st), "riggs_raster_preprocess")
    s = _Saved()
    if arena is None or arena.last_R < 0:
        R = int(counters[0].item())  # device->host sync, as upstream does
        cap = R
        if arena is not None:
            binning = arena.ensure(int(R * arena.growth) + 1, N, H, W, dev)
            cap = arena.capacity
            arena.last_R = R
        else:
            binning = torch.empty(lib.riggs_raster_binning_bytes(cap, N, H, W), dtype=torch.uint8, device=dev)
            L.check(lib.riggs_raster_binning_reset_history(binning.data_ptr(), cap, N, H, W, st), "riggs_raster_binning_reset_history")
        s.R = R
    else:
        arena.resolve(block=True)
        # (inside a hipGraph capture an arena that still has a third of the head-room is kept — its owner topped it up before
        # capturing, RasterArena.top_up — so that the captured frame neither allocates nor resets the walk history; an eagerly
        # issued frame gets the full head-room back whenever the count reached a new maximum: a scene under training can grow by
        # 10 % from one frame to the next)
        minimum = int(arena.last_R * (1.0 + (arena.growth - 1.0) / 3.0)) + 1 if torch.cuda.is_current_stream_capturing() else None
        binning = arena.ensure(int(arena.last_R * arena.growth) + 1, N, H, W, dev, minimum=minimum)
        cap = arena.capacity
        s.R = None  # unknown until counters are read
    L.check(lib.riggs_raster_render(C.byref(cfg), geom.data_ptr(), binning.data_ptr(), cap, binning.numel(), img.data_ptr(),
                                    color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), counters.data_ptr(), st),
            "riggs_raster_render")
    if arena is not None and s.R is None:
        arena._post(counters, cap)
    s.cfg, s.keep, s.geom, s.img, s.binning, s.cap, s.radii, s.counters = cfg, keep, geom, img, binning, cap, radii, counters
    s.N, s.H, s.W, s.M = N, H, W, M
    return color, radii, depth, alpha, s


def arena_check(s: _Saved, arena: RasterArena) -> bool:
    """Reads the instance count of a frame rendered through an arena.  Returns False when the
    arena overflowed (the caller must re-render)."""
    c = s.counters[:2].tolist()
    s.R = int(c[0]) & 0xFFFFFFFF
    arena.last_R = s.R
    return c[1] == 0


def rasterize_backward(s: _Saved, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                       d_xyz, d_rotation, grad_color, grad_depth, grad_alpha, d_scaling=None,
                       want_d_scaling_grad=False, shs_rest=None, sparse_rows=False):
    """``sparse_rows``: the frame's arena belongs to a captured frame whose owner asked for sparse gradient rows
    (RasterArena.sparse_grad_rows)."""
    lib = L.lib()
    N, M, dev = s.N, s.M, means3D.device
    f32 = dict(dtype=torch.float32, device=dev)
    cfg = s.cfg
    # (grad_out: the parameter's slice of a registered flat gradient bucket — riggs_amd/dist.py — else a fresh buffer)
    from .dist import grad_out
    g_means3D = grad_out(means3D, (N, 3))
    g_means2D = torch.empty(N, 3, **f32)
    g_sh = grad_out(shs, (N, shs.shape[1], 3)) if shs is not None else None
    g_sh_rest = grad_out(shs_rest, (N, shs_rest.shape[1], 3)) if shs_rest is not None else None
    g_colors = torch.empty(N, 3, **f32) if colors_precomp is not None else None
    g_opac = grad_out(opacities, (N, 1))
    iso = bool(cfg.glue and cfg.isotropic)
    g_scales = grad_out(scales, (N, 1 if iso else 3)) if scales is not None else None
    g_rots = grad_out(rotations, (N, 4)) if rotations is not None else None
    g_cov = torch.empty(N, 6, **f32) if cov3D_precomp is not None else None
    g_dscaling = torch.empty(N, 3, **f32) if (d_scaling is not None and want_d_scaling_grad) else None
    # sparse zero-fill (riggs_raster_cfg.sparse_zero): only inside a hipGraph capture whose owner asked for it
    # (GraphedFrame(sparse_grad_rows=True): static gradient buffers, zero-filled together with the workspace after the
    # capture), with the fused glue (the outputs ARE the parameters' gradient buffers, not autograd intermediates whose
    # memory the graph's pool may hand to a later allocation) and without the optional outputs
    sparse = bool(sparse_rows and cfg.glue and not cfg.deterministic and g_colors is None and g_cov is None
                  and g_dscaling is None and torch.cuda.is_current_stream_capturing())
    if sparse:
        # ... and every output must be PERSISTENT memory that nothing else is ever placed in: a tensor allocated inside the
        # capture shares its block of the graph's pool with earlier intermediates of the same replay (measured: dL/d_xyz
        # landed on the block the forward's d_xyz had just left — the rows not rewritten then hold d_xyz).  Bucket slices
        # (riggs_amd.dist) qualify; the screen-space gradient gets a buffer of its own below.
        from .dist import in_bucket
        pairs = [(means3D, g_means3D), (shs, g_sh), (shs_rest, g_sh_rest), (opacities, g_opac), (scales, g_scales), (rotations, g_rots)]
        sparse = all(in_bucket(p, g) for p, g in pairs if g is not None)
    if sparse_rows and cfg.glue:
        # (the buffer is created by the owner's EAGER warm-up frames on this stream — outside any graph pool — and handed
        # out as a fresh view per call, so that autograd adopts it as viewspace_points.grad instead of cloning it)
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), L.stream_ptr(), N)
        if key not in _MEANS2D and not torch.cuda.is_current_stream_capturing():
            if len(_MEANS2D) >= 8:
                _MEANS2D.pop(next(iter(_MEANS2D)))
            _MEANS2D[key] = torch.zeros(N, 3, **f32)
        if sparse and key in _MEANS2D:
            g_means2D = _MEANS2D[key].view(N, 3)
        else:
            sparse = False
    cfg.sparse_zero = 1 if sparse else 0
    if cfg.deterministic:  # (its accumulators are overwritten by the ordered sum: nothing to keep zeroed)
        ws = torch.zeros(lib.riggs_raster_backward_workspace_bytes_ordered(N, s.cap), dtype=torch.uint8, device=dev)
    else:
        ws = _backward_workspace(lib.riggs_raster_backward_workspace_bytes(N), dev, N)
    if grad_color is None:  # a loss on depth / alpha only (set_materialize_grads(False) hands None for the unused output)
        grad_color = torch.zeros(3, s.H, s.W, **f32)
    gc = L.require_cuda_f32("grad_color", grad_color, (3, s.H, s.W))
    gd = L.require_cuda_f32("grad_depth", grad_depth) if grad_depth is not None else None
    ga = L.require_cuda_f32("grad_alpha", grad_alpha) if grad_alpha is not None else None
    try:
        L.check(lib.riggs_raster_backward(
            C.byref(cfg), L.ptr(means3D), L.ptr(shs), L.ptr(shs_rest), L.ptr(colors_precomp), L.ptr(opacities), L.ptr(scales),
            L.ptr(rotations), L.ptr(cov3D_precomp), L.ptr(d_xyz), L.ptr(d_rotation), L.ptr(d_scaling), s.radii.data_ptr(),
            s.geom.data_ptr(), s.binning.data_ptr(), s.cap, s.img.data_ptr(), s.counters.data_ptr(), gc.data_ptr(),
            L.ptr(gd), L.ptr(ga), ws.data_ptr(), g_means3D.data_ptr(), g_means2D.data_ptr(), L.ptr(g_sh),
            L.ptr(g_colors), g_opac.data_ptr(), L.ptr(g_scales), L.ptr(g_rots), L.ptr(g_cov), L.ptr(g_dscaling),
            L.ptr(g_sh_rest), L.stream_ptr()),
            "riggs_raster_backward")
    except Exception:
        _WORKSPACES.clear()  # a failed call may leave the accumulators dirty: the next one starts from fresh zeros
        raise
    _LAST_WORKSPACE[:] = [ws, N]
    # (addresses, not tensors: an extra reference would make autograd's AccumulateGrad clone the gradient instead of
    # adopting the kernel's buffer as the parameter's .grad)
    _LAST_SPARSE_OUTPUTS[:] = [t.data_ptr() for t in (g_means3D, g_means2D, g_sh, g_sh_rest, g_opac, g_scales, g_rots) if t is not None] if sparse else []
    if shs_rest is not None:
        g_sh = (g_sh, g_sh_rest)
    return g_means3D, g_means2D, g_sh, g_colors, g_opac, g_scales, g_rots, g_cov, g_dscaling


_WORKSPACES = {}
_LAST_WORKSPACE = [None, 0]
_MEANS2D = {}              # (device, stream, N) -> persistent screen-space gradient buffer of the sparse mode
_LAST_SPARSE_OUTPUTS = []  # addresses of the gradient buffers of the most recent backward that ran with cfg.sparse_zero


def mark_all_rows(workspace: torch.Tensor, N: int):
    """Tell the next sparse backward that every gradient row may hold something (after a writer that does not keep the row
    list, e.g. a dense all-reduce in place): it then rewrites all of them."""
    import ctypes as C
    off, nbytes = C.c_size_t(), C.c_size_t()
    L.check(L.lib().riggs_raster_backward_workspace_rows(N, C.byref(off), C.byref(nbytes)), "riggs_raster_backward_workspace_rows")
    workspace[off.value:off.value + nbytes.value].fill_(0xFF)


def last_backward_workspace():
    """(workspace, N) of the most recent ``rasterize_backward``: behind its accumulators the call leaves which Gaussians
    received a gradient (include/riggs_hip.h: riggs_grad_rows_pack reads it; riggs_amd.dist.SparseRowExchange)."""
    if _LAST_WORKSPACE[0] is None:
        raise RuntimeError("no rasterizer backward has run yet")
    return _LAST_WORKSPACE[0], _LAST_WORKSPACE[1]


def _backward_workspace(nbytes: int, dev, N: int) -> torch.Tensor:
    """The per-Gaussian gradient accumulators of the compositing backward (include/riggs_hip.h: `workspace`): zeroed ONCE
    here, then self-cleaning — the kernels leave them all zero — so a frame pays no fill pass.  One buffer per (device,
    stream, N): a backward call uses it from its first to its last kernel on one stream, and the row list behind the
    accumulators sits at an offset that depends on N (a buffer shared between sizes would find it inside its accumulators)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), L.stream_ptr(), N)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        if len(_WORKSPACES) >= 8:  # (a training run has one or two sizes alive; densification retires the old ones)
            _WORKSPACES.pop(next(iter(_WORKSPACES)))
        ws = _WORKSPACES[key] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    return ws


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings):
        ctx.set_materialize_grads(False)
        color, radii, depth, alpha, s = rasterize_forward(settings, means3D, sh, colors_precomp, opacities, scales,
                                                          rotations, cov3Ds_precomp)
        ctx.s = s
        ctx.save_for_backward(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp = ctx.saved_tensors
        g = rasterize_backward(ctx.s, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, None,
                               None, grad_color, grad_depth, grad_alpha)
        g_means3D, g_means2D, g_sh, g_colors, g_opac, g_scales, g_rots, g_cov, _ = g
        return g_means3D, g_means2D, g_sh, g_colors, g_opac, g_scales, g_rots, g_cov, None


def _prep(name, t, shape):
    if t is None or (isinstance(t, torch.Tensor) and t.numel() == 0 and shape[0] != 0 and t.dim() == 1):
        return None
    return L.require_cuda_f32(name, t, shape)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Frustum test of upstream's markVisible: view-space z > 0.2."""
        with torch.no_grad():
            V = self.raster_settings.viewmatrix
            z = positions @ V[:3, 2] + V[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        N = means3D.shape[0]
        means3D = L.require_cuda_f32("means3D", means3D, (N, 3))
        if means2D is None:
            means2D = torch.zeros_like(means3D)
        opacities = L.require_cuda_f32("opacities", opacities.reshape(N, 1), (N, 1))
        shs = L.require_cuda_f32("shs", shs, (N, None, 3)) if shs is not None else None
        colors_precomp = L.require_cuda_f32("colors_precomp", colors_precomp, (N, 3)) if colors_precomp is not None else None
        scales = L.require_cuda_f32("scales", scales, (N, 3)) if scales is not None else None
        rotations = L.require_cuda_f32("rotations", rotations, (N, 4)) if rotations is not None else None
        cov3D_precomp = L.require_cuda_f32("cov3D_precomp", cov3D_precomp, (N, 6)) if cov3D_precomp is not None else None
        return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                         cov3D_precomp, self.raster_settings)


# ---- debugging / test access to the opaque arenas --------------------------------------
def saved_views(s: _Saved):
    """Typed views into the geometry / image / binning arenas of a forward call (tests)."""
    lib = L.lib()
    go = (C.c_size_t * L.GEOM_NFIELDS)()
    io = (C.c_size_t * L.IMG_NFIELDS)()
    bo = (C.c_size_t * L.BIN_NFIELDS)()
    lib.riggs_raster_geom_layout(s.N, go)
    lib.riggs_raster_image_layout(s.H, s.W, io)
    lib.riggs_raster_binning_layout(s.cap, s.N, s.H, s.W, bo)
    N, H, W = s.N, s.H, s.W
    T = ((W + 15) // 16) * ((H + 15) // 16)
    if s.R is None:
        s.R = int(s.counters[0].item())

    def view(buf, off, nbytes, dtype, shape):
        return buf[off:off + nbytes].view(dtype).reshape(shape)
    out = {
        "xyd": view(s.geom, go[L.GEOM_XYD], N * 16, torch.float32, (N, 4)),
        "conic_o": view(s.geom, go[L.GEOM_CONIC_O], N * 16, torch.float32, (N, 4)),
        "rgb": view(s.geom, go[L.GEOM_RGB], N * 16, torch.float32, (N, 4)),
        "cov3D": view(s.geom, go[L.GEOM_COV3D], N * 24, torch.float32, (N, 6)),
        "clamped": view(s.geom, go[L.GEOM_CLAMPED], N, torch.uint8, (N,)),
        "tiles_touched": view(s.geom, go[L.GEOM_TILES], N * 4, torch.int32, (N,)),
        "rect": view(s.geom, go[L.GEOM_RECT], N * 8, torch.int16, (N, 4)),
        "depth_order": view(s.geom, go[L.GEOM_DEPTH_ORDER], N * 4, torch.int32, (N,)),
        "final_T": view(s.img, io[L.IMG_FINAL_T], H * W * 4, torch.float32, (H, W)),
        "n_contrib": view(s.img, io[L.IMG_N_CONTRIB], H * W * 4, torch.int32, (H, W)),
        "ranges": view(s.img, io[L.IMG_RANGES], T * 8, torch.int32, (T, 2)),
        "fwd_ctr": view(s.img, io[L.IMG_FWD_CTR], 12, torch.int32, (3,)),
        "point_list": view(s.binning, bo[L.BIN_POINT_LIST], s.R * 4, torch.int32, (s.R,)),
        "tile_keys": view(s.binning, bo[L.BIN_TILE_KEYS], s.R * 4, torch.int32, (s.R,)),
        "R": s.R,
    }
    if not s.cfg.debug:
        # the counting-sort binning stores the per-instance tile id only with settings.debug; it is implied by
        # the tile ranges (instance p belongs to the tile whose range holds p)
        rg = out["ranges"].long()
        lens = (rg[:, 1] - rg[:, 0]).clamp(min=0)
        out["tile_keys"] = torch.repeat_interleave(torch.arange(T, device=rg.device), lens).to(torch.int32)[:s.R]
    return out
