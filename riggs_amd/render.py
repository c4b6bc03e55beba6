"""Host-side mirror of ``render()`` (/root/reference/gaussian_renderer/__init__.py:37-151).

Two paths, same signature and return dict:
  * general path — the reference's own sequence of activations, then the drop-in
    ``GaussianRasterizer`` (every optional branch of the reference works);
  * fused path (``fused=True`` and the default branch: SH colours from the rasterizer,
    scales/rotations, tensor or 0.0 residuals) — raw parameters go straight to the HIP
    preprocess kernel which applies sigmoid / exp / normalize(_rotation + d_rotation) /
    xyz + d_xyz in registers, and the backward kernel applies their chain rule, so none
    of those (N,k) intermediates ever round-trips HBM.
"""
from __future__ import annotations

import math

import torch

from . import _lib as L
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, RasterArena, rasterize_forward,
                         rasterize_backward, arena_check)


def quaternion_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    o = torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)
    return torch.where(o[..., 0:1] < 0, -o, o)


class RenderPkg(dict):
    """The dict ``render`` returns.  ``visibility_filter`` (= ``radii > 0``, gaussian_renderer/__init__.py:147) is
    an elementwise launch that training does not need every frame: it is evaluated on first access.  With
    ``cache=False`` (static buffers of a replayed hipGraph) it is re-evaluated on every access."""

    def __init__(self, *a, cache=True, **k):
        super().__init__(*a, **k)
        self._cache = cache
        dict.setdefault(self, "visibility_filter", None)

    def _vis(self):
        v = dict.__getitem__(self, "visibility_filter")
        if v is None:
            v = dict.__getitem__(self, "radii") > 0
            if self._cache:
                dict.__setitem__(self, "visibility_filter", v)
        return v

    def __getitem__(self, key):
        return self._vis() if key == "visibility_filter" else dict.__getitem__(self, key)

    def get(self, key, default=None):
        return self._vis() if key == "visibility_filter" else dict.get(self, key, default)

    def items(self):
        return [(k, self[k]) for k in dict.keys(self)]

    def values(self):
        return [self[k] for k in dict.keys(self)]


_ZERO_POINTS = {}


def _zero_points(xyz):
    """A fresh autograd leaf of zeros shaped like ``xyz`` (the reference's ``screenspace_points``) that aliases
    one cached zero buffer: nothing ever writes its values, so no fill kernel per frame."""
    key = (xyz.device, xyz.shape[0])
    z = _ZERO_POINTS.get(key)
    if z is None:
        _ZERO_POINTS.clear()
        z = _ZERO_POINTS[key] = torch.zeros(xyz.shape[0], 3, dtype=torch.float32, device=xyz.device)
    return z.detach().requires_grad_(True)


class _FusedGlueRaster(torch.autograd.Function):
    """render glue + rasterizer as ONE autograd node over the raw Gaussian parameters."""

    @staticmethod
    def forward(ctx, xyz, means2D, f_dc, f_rest, opacity, scaling, rotation, d_xyz, d_rot, d_scaling, settings,
                isotropic, arena):
        ctx.set_materialize_grads(False)
        N = xyz.shape[0]
        f_dc = L.require_cuda_f32("_features_dc", f_dc, (N, 1, 3))      # read in place: no torch.cat, the kernel
        f_rest = L.require_cuda_f32("_features_rest", f_rest, (N, None, 3))  # stages both arrays through LDS
        xyz = L.require_cuda_f32("_xyz", xyz, (N, 3))
        opacity = L.require_cuda_f32("_opacity", opacity, (N, 1))
        scaling = L.require_cuda_f32("_scaling", scaling, (N, 1 if isotropic else 3))
        rotation = L.require_cuda_f32("_rotation", rotation, (N, 4))
        d_xyz = L.require_cuda_f32("d_xyz", d_xyz, (N, 3)) if d_xyz is not None else None
        d_rot = L.require_cuda_f32("d_rotation", d_rot, (N, 4)) if d_rot is not None else None
        d_scaling = L.require_cuda_f32("d_scaling", d_scaling, (N, 3)) if d_scaling is not None else None
        out = rasterize_forward(settings, xyz, f_dc, None, opacity, scaling, rotation, None, d_xyz=d_xyz,
                                d_rotation=d_rot, d_scaling=d_scaling, glue=True, isotropic=isotropic, arena=arena,
                                shs_rest=f_rest)
        color, radii, depth, alpha, s = out
        ctx.s, ctx.arena, ctx.settings, ctx.isotropic = s, arena, settings, isotropic
        ctx.save_for_backward(xyz, f_dc, f_rest, opacity, scaling, rotation, d_xyz, d_rot, d_scaling)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_depth, g_alpha):
        xyz, f_dc, f_rest, opacity, scaling, rotation, d_xyz, d_rot, d_scaling = ctx.saved_tensors
        s = ctx.s
        if ctx.arena is not None:
            ctx.arena.resolve(block=False)  # raises if the forward of this frame is known to have overflowed
        need_ds = d_scaling is not None and ctx.needs_input_grad[9]
        g = rasterize_backward(s, xyz, f_dc, None, opacity, scaling, rotation, None, d_xyz, d_rot, g_color, g_depth,
                               g_alpha, d_scaling=d_scaling, want_d_scaling_grad=need_ds, shs_rest=f_rest,
                               sparse_rows=bool(ctx.arena is not None and ctx.arena.sparse_grad_rows))
        g_means3D, g_means2D, (g_dc, g_rest), _, g_opac, g_scales, g_rots, _, g_ds = g
        # dL/d(d_xyz) == dL/dxyz and dL/d(d_rotation) == dL/d_rotation: hand the residual branches an alias (a
        # second tensor object on the same storage) so that AccumulateGrad can adopt the parameter gradients
        # instead of cloning them because they are referenced twice
        return (g_means3D, g_means2D, g_dc, g_rest, g_opac, g_scales, g_rots,
                g_means3D.detach() if d_xyz is not None else None, g_rots.detach() if d_rot is not None else None,
                g_ds, None, None, None)


def _extension_frame(settings, pc, arena, dx, dr, ds, scaling, iso, screenspace_points):
    """The default branch through the PyTorch extension's node (csrc_torch/riggs_torch.cpp: torch.ops.riggs.glue_raster) when the call
    is the plain eager training frame: an arena that already knows the previous frame's instance count (so that no host read is
    needed), no registered gradient bucket, no capture, no ordered backward, no sparse rows, no ``pipe.debug``.  The arena policy —
    sizing, the asynchronous read-back of the count, overflow reporting — stays here, with ``RasterArena``.  None = take the ctypes
    node."""
    from . import _torch_ext as TX
    from . import rasterizer as R
    from .dist import _SLICES, _entry
    if (arena is None or arena.last_R < 0 or arena.sparse_grad_rows or R.ORDERED_BACKWARD or settings.debug
            or not TX.active() or not pc._xyz.is_cuda):
        return None
    if _SLICES and any(_entry(p) is not None for p in (pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation)):
        return None  # (these parameters' gradients belong into a registered flat bucket: the ctypes node writes them there)
    N, dev = pc._xyz.shape[0], pc._xyz.device
    H, W = int(settings.image_height), int(settings.image_width)
    arena.resolve(block=True)
    binning = arena.ensure(int(arena.last_R * arena.growth) + 1, N, H, W, dev)
    cap = arena.capacity
    ws = R._backward_workspace(L.lib().riggs_raster_backward_workspace_bytes(N), dev, N)
    bg = settings.bg if settings.bg.device == dev else settings.bg.to(dev)
    color, radii, depth, alpha, counters = torch.ops.riggs.glue_raster(
        pc._xyz, screenspace_points, pc._features_dc, pc._features_rest, pc._opacity, scaling, pc._rotation, dx, dr, ds, bg,
        settings.viewmatrix, settings.projmatrix, settings.campos, binning, ws, cap, H, W, float(settings.tanfovx),
        float(settings.tanfovy), float(settings.scale_modifier), int(settings.sh_degree), False, iso, bool(arena.tight_lists))
    arena._post(counters, cap)
    R._LAST_WORKSPACE[:] = [ws, N]
    R._LAST_SPARSE_OUTPUTS[:] = []
    return color, radii, depth, alpha


def _is_zero_scalar(v):
    return (not isinstance(v, torch.Tensor)) and float(v) == 0.0


def render(viewpoint_camera, pc, pipe, bg_color, d_xyz, d_rotation, d_scaling, d_opacity=None, d_color=None,
           scaling_modifier=1.0, override_color=None, random_bg_color=False, render_motion=False, detach_xyz=False,
           detach_scale=False, detach_rot=False, detach_opacity=False, d_rot_as_res=True, scale_const=None,
           d_rotation_bias=None, force_visible=False, fused=True, arena: RasterArena = None):
    """Same contract as the reference ``render`` (returns the same dict).  ``fused`` / ``arena`` are additions."""
    xyz = pc.get_xyz
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    bg = bg_color if not random_bg_color else torch.rand_like(bg_color)
    settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)

    default_branch = (not pipe.compute_cov3D_python and not pipe.convert_SHs_python and not render_motion
                      and override_color is None and scale_const is None and d_opacity is None
                      and d_rotation_bias is None and (d_color is None or type(d_color) is float)
                      and not (detach_xyz or detach_scale or detach_rot or detach_opacity))
    if fused and default_branch:
        screenspace_points = _zero_points(xyz)  # leaf: .grad is populated by autograd
        dx = None if _is_zero_scalar(d_xyz) else d_xyz
        dr = None if _is_zero_scalar(d_rotation) else d_rotation
        ds = None if _is_zero_scalar(d_scaling) else d_scaling
        iso = bool(getattr(pc, "use_isotropic_gs", False))
        scaling = pc._scaling[..., :1] if iso else pc._scaling
        fast = _extension_frame(settings, pc, arena, dx, dr, ds, scaling, iso, screenspace_points)
        if fast is not None:
            color, radii, depth, alpha = fast
        else:
            color, radii, depth, alpha = _FusedGlueRaster.apply(
                pc._xyz, screenspace_points, pc._features_dc, pc._features_rest, pc._opacity, scaling, pc._rotation,
                dx, dr, ds, settings, iso, arena)
        return RenderPkg({"render": color, "viewspace_points": screenspace_points, "visibility_filter": None,
                          "radii": radii, "depth": depth, "alpha": alpha, "bg_color": bg})

    # ---- general path: every optional branch of the reference's render(), resolved by three small helpers and
    # handed to the drop-in GaussianRasterizer (the activations are torch ops here; the rasterizer is HIP)
    screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0  # non-leaf, as the reference builds it
    screenspace_points.retain_grad()
    cut = lambda t, flag: t.detach() if (flag and t is not None) else t  # noqa: E731
    means3D = cut(xyz + d_xyz, detach_xyz)
    opacity = _general_opacity(pc, d_opacity, scale_const is not None)
    shape = _general_shape(pc, pipe, scaling_modifier, d_rotation, d_scaling, d_rotation_bias, scale_const)
    shape = {k: cut(v, (detach_rot or detach_scale) if k == "cov3D_precomp" else
                    (detach_rot if k == "rotations" else detach_scale)) for k, v in shape.items()}
    colour = _general_colour(pc, pipe, viewpoint_camera, xyz, d_color, override_color, render_motion)
    image, radii, depth, alpha = GaussianRasterizer(raster_settings=settings)(
        means3D=means3D, means2D=screenspace_points, opacities=cut(opacity, detach_opacity), **shape, **colour)
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
            "depth": depth, "alpha": alpha, "bg_color": bg}


def _general_opacity(pc, d_opacity, constant_scale):
    """gaussian_renderer/__init__.py:76-82: opaque splats when a constant scale is forced, else sigmoid(_opacity) (+ residual)."""
    base = pc.get_opacity
    if constant_scale:
        return torch.ones_like(base)
    return base if d_opacity is None else base + d_opacity


def _general_shape(pc, pipe, scaling_modifier, d_rotation, d_scaling, d_rotation_bias, scale_const):
    """Either the python 3-D covariance (:86-87) or the (scales, rotations) pair (:89-92, :129-130)."""
    if pipe.compute_cov3D_python:
        dr = d_rotation if isinstance(d_rotation, torch.Tensor) else None
        return {"scales": None, "rotations": None,
                "cov3D_precomp": pc.get_covariance(scaling_modifier, d_rotation=dr, gs_rot_bias=d_rotation_bias)}
    rotations = pc.get_rotation_bias(d_rotation)
    if d_rotation_bias is not None:
        rotations = quaternion_multiply(d_rotation_bias, rotations)
    scales = pc.get_scaling + d_scaling
    if scale_const is not None:
        scales = torch.full_like(scales, float(scale_const))
    return {"scales": scales, "rotations": rotations, "cov3D_precomp": None}


def _general_colour(pc, pipe, cam, xyz, d_color, override_color, render_motion):
    """Where the splat colour comes from (:94-116): the motion-mask visualisation, a caller-supplied colour, SH evaluated
    in python, or the SH coefficients themselves (evaluated by the rasterizer)."""
    if render_motion:
        mm = pc.motion_mask
        return {"shs": None, "colors_precomp": torch.cat([mm, torch.zeros_like(mm), 1 - mm], dim=-1)}
    if override_color is not None:
        return {"shs": None, "colors_precomp": override_color}
    feats = pc.get_features
    if isinstance(d_color, torch.Tensor):
        feats = torch.cat([feats[:, :1] + d_color[:, None], feats[:, 1:]], dim=1)
    if not pipe.convert_SHs_python:
        return {"shs": feats, "colors_precomp": None}
    from .sh import eval_sh
    view_dir = torch.nn.functional.normalize(xyz - cam.camera_center[None], dim=1, eps=0.0)
    rgb = eval_sh(pc.active_sh_degree, feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2), view_dir)
    return {"shs": None, "colors_precomp": torch.clamp_min(rgb + 0.5, 0.0)}
