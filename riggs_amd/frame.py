"""The whole per-frame hot path as ONE autograd node over ONE C call per direction (riggs_frame_forward / riggs_frame_backward,
include/riggs_hip.h): what train_rig.py:535-554 issues as ``skeleton.step()`` (:411) + ``render()`` (:488) + ``loss.backward()``.

``deform_render`` returns the reference ``render``'s dict (render, viewspace_points, visibility_filter, radii, depth, alpha,
bg_color) plus the deformation's outputs a trainer reads (d_nodes, local_rotation, global_trans — differentiable: the
projection loss and the pose regularisers hang on them — and d_xyz, d_rotation, d_scaling as values).  Same kernels and results
as ``SkeletonWarp.forward`` + ``render(fused=True)``; what changes is the host: two autograd nodes and seven ctypes crossings
become one and two (an eagerly issued frame at the bench workload is host-bound otherwise: bench.py ``eager_api``).  Frames the
entry does not cover — the first frame of an arena (its size is not known yet), the MLP heads, top-K skinning, the optional
branches of ``render`` — go through the separate calls, so the function can replace them unconditionally.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib as L
from .rasterizer import GaussianRasterizationSettings, RasterArena, _backward_workspace, _cfg, _LAST_WORKSPACE
from .render import RenderPkg, _zero_points, render
from .skeleton import _PoseMLPFn


def _req(name, t, shape):
    """L.require_cuda_f32 with its common case inline (a contiguous float32 device tensor of the expected leading size): the
    frame entry exists to take host time out of the eager frame, and a dozen full checks per frame are 25 us of it."""
    if t.is_cuda and t.dtype is torch.float32 and t.is_contiguous() and t.shape[0] == shape[0] and t.dim() == len(shape):
        return t
    return L.require_cuda_f32(name, t, shape)


_SIZES = {}  # (N, H, W, PoseMLP shape) -> sizes the library reports for them


class _FrameFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, rho, mask, xyz, means2D, f_dc, f_rest, opacity, scaling, rotation, spec, *params):
        ctx.set_materialize_grads(False)
        sw, settings, arena, isotropic = spec
        lib, dev = L.lib(), xyz.device
        pn = sw.pose_net
        depth, width = len(pn.net), pn.net[0].out_features
        N, J = xyz.shape[0], sw.nodes.shape[0]
        H, W = int(settings.image_height), int(settings.image_width)
        f32 = dict(dtype=torch.float32, device=dev)
        params = [p.contiguous() for p in params]
        xyz = _req("_xyz", xyz, (N, 3))
        f_dc = _req("_features_dc", f_dc, (N, 1, 3))
        f_rest = _req("_features_rest", f_rest, (N, None, 3))
        opacity = _req("_opacity", opacity, (N, 1))
        scaling = _req("_scaling", scaling, (N, 1 if isotropic else 3))
        rotation = _req("_rotation", rotation, (N, 4))
        rho = _req("_node_radius", rho, (J,))
        mflat = None if mask is None else _req("motion_mask", mask.reshape(-1), (N,))
        joints, par = sw._joints(), sw._parents_dev(dev)
        sync = pn._hip_sync
        if sync.device != dev or sync.numel() * 4 < lib.riggs_pose_mlp_sync_bytes(depth, width):
            sync = None
        else:
            pn.watch()
        keep = []
        cfg = _cfg(settings, N, f_dc.shape[1] + f_rest.shape[1], True, isotropic, keep, arena.tight_lists)
        # outputs and saved state: ONE float allocation for the per-joint arrays, the PoseMLP's activations, the residuals and
        # the counters, ONE byte allocation for the two rasterizer arenas (ten torch.empty calls were 29 us of an eager frame;
        # every piece starts on a 256-byte boundary)
        def up(n):
            return (n + 63) & ~63
        skey = (N, H, W, depth, width, pn.multires)
        sizes = _SIZES.get(skey)
        if sizes is None:
            if len(_SIZES) > 64:
                _SIZES.clear()
            sizes = _SIZES[skey] = (lib.riggs_pose_mlp_acts_floats(depth, width, pn.multires), lib.riggs_raster_geom_bytes(N),
                                    lib.riggs_raster_image_bytes(H, W))
        n_small, n_acts = J * 23 + 4, sizes[0]
        o_acts = up(n_small)
        o_dx = o_acts + up(n_acts)
        o_dr = o_dx + up(3 * N)
        o_cnt = o_dr + up(4 * N)
        o_rad = o_cnt + 64
        fbuf = torch.empty(o_rad + up(N), **f32)
        small, acts = fbuf[:n_small], fbuf[o_acts:o_acts + n_acts]
        local_rot, transforms = small[:J * 4].view(J, 4), small[J * 4:J * 16].view(J, 12)
        node_rot, d_nodes, global_trans = small[J * 16:J * 20].view(J, 4), small[J * 20:J * 23].view(J, 3), small[J * 23:J * 23 + 3]
        d_xyz, d_rot = fbuf[o_dx:o_dx + 3 * N].view(N, 3), fbuf[o_dr:o_dr + 4 * N].view(N, 4)
        ibuf = fbuf.view(torch.int32)
        counters, radii = ibuf[o_cnt:o_cnt + 4], ibuf[o_rad:o_rad + N]
        n_geom = (sizes[1] + 255) & ~255
        bbuf = torch.empty(n_geom + sizes[2], dtype=torch.uint8, device=dev)
        geom, img = bbuf[:n_geom], bbuf[n_geom:]
        out = torch.empty(5, H, W, **f32)
        color, depth_img, alpha = out[:3], out[3:4], out[4:5]
        arena.resolve(block=True)
        binning = arena.ensure(int(arena.last_R * arena.growth) + 1, N, H, W, dev)
        Wp, bp = _PoseMLPFn._ptrs(params, depth)
        h = params[2 * depth:]
        fr = L.Frame()
        fr.depth, fr.width, fr.multires, fr.skip, fr.n_rot = depth, width, pn.multires, pn.skips[0], h[0].shape[0]
        fr.weights, fr.biases = C.cast(Wp, C.c_void_p), C.cast(bp, C.c_void_p)
        fr.W_rot, fr.b_rot, fr.W_tr, fr.b_tr = h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr(), h[3].data_ptr()
        fr.t, fr.rot_bias4, fr.sync_state, fr.acts = t.data_ptr(), L.ptr(sw._rot_bias), L.ptr(sync), acts.data_ptr()
        fr.local_rot, fr.global_trans = local_rot.data_ptr(), global_trans.data_ptr()
        fr.num_joints, fr.K = J, sw.K
        fr.joints, fr.parents, fr.node_radius_log = joints.data_ptr(), par.data_ptr(), rho.data_ptr()
        fr.motion_mask, fr.weight_mod = L.ptr(mflat), None
        fr.transforms, fr.node_rot, fr.d_nodes = transforms.data_ptr(), node_rot.data_ptr(), d_nodes.data_ptr()
        fr.d_xyz, fr.d_rotation = d_xyz.data_ptr(), d_rot.data_ptr()
        fr.cfg = cfg
        fr.xyz, fr.features_dc, fr.features_rest = xyz.data_ptr(), f_dc.data_ptr(), f_rest.data_ptr()
        fr.opacity, fr.scaling, fr.rotation, fr.d_scaling = opacity.data_ptr(), scaling.data_ptr(), rotation.data_ptr(), None
        fr.geom, fr.radii, fr.counters = geom.data_ptr(), radii.data_ptr(), counters.data_ptr()
        fr.binning, fr.instance_capacity, fr.binning_bytes, fr.image_state = binning.data_ptr(), arena.capacity, binning.numel(), img.data_ptr()
        fr.out_color, fr.out_depth, fr.out_alpha = color.data_ptr(), depth_img.data_ptr(), alpha.data_ptr()
        L.check(lib.riggs_frame_forward(C.byref(fr), L.stream_ptr()), "riggs_frame_forward")
        arena._post(counters, arena.capacity)
        ctx.save_for_backward(t, rho, mflat, xyz, f_dc, f_rest, opacity, scaling, rotation, *params)
        ctx.fr = fr
        ctx.keep = (keep, Wp, bp, small, acts, d_xyz, d_rot, geom, img, radii, counters, binning, joints, par, sync, out)
        ctx.sizes = (N, J, depth, width, pn.multires, isotropic, mask.shape if mask is not None else None)
        ctx.arena = arena
        ctx.mark_non_differentiable(radii, d_xyz, d_rot)
        return color, radii, depth_img, alpha, d_nodes, local_rot, global_trans, d_xyz, d_rot

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_depth, g_alpha, g_nodes, g_local_rot, g_global_trans, _g_dx, _g_dr):
        t, rho, mflat, xyz, f_dc, f_rest, opacity, scaling, rotation, *params = ctx.saved_tensors
        N, J, depth, width, multires, isotropic, mask_shape = ctx.sizes
        fr, lib, dev = ctx.fr, L.lib(), xyz.device
        f32 = dict(dtype=torch.float32, device=dev)
        ctx.arena.resolve(block=False)  # raises if this frame's forward is known to have overflowed its arena
        from .dist import grad_out, grad_out_flat
        H, W = fr.cfg.image_height, fr.cfg.image_width
        if g_color is None:
            g_color = torch.zeros(3, H, W, **f32)
        gc = L.require_cuda_f32("grad_color", g_color, (3, H, W))
        gd = L.require_cuda_f32("grad_depth", g_depth) if g_depth is not None else None
        ga = L.require_cuda_f32("grad_alpha", g_alpha) if g_alpha is not None else None
        g_xyz, g_m2d = grad_out(xyz, (N, 3)), torch.empty(N, 3, **f32)
        g_dc, g_rest = grad_out(f_dc, tuple(f_dc.shape)), grad_out(f_rest, tuple(f_rest.shape))
        g_op, g_sc, g_rot = grad_out(opacity, (N, 1)), grad_out(scaling, (N, 1 if isotropic else 3)), grad_out(rotation, (N, 4))
        drho = grad_out(rho, (J,))
        need_mask = mflat is not None and ctx.needs_input_grad[2]
        dmask = torch.empty(N, **f32) if need_mask else None
        small = torch.empty(J * 16 + 8, **f32)
        dG, dq, dgt_s, dgt = small[:J * 12], small[J * 12:J * 16].view(J, 4), small[J * 16:J * 16 + 3], small[J * 16 + 4:J * 16 + 7]
        flat = grad_out_flat(params)
        ws = _backward_workspace(lib.riggs_raster_backward_workspace_bytes(N), dev, N)
        lws = torch.empty(lib.riggs_lbs_backward_workspace_bytes(N, J), dtype=torch.uint8, device=dev)
        dzs = torch.empty(lib.riggs_pose_mlp_backward_workspace_floats(depth, width, multires), **f32)
        gn = None if g_nodes is None else L.require_cuda_f32("g_nodes", g_nodes, (J, 3))
        gq = None if g_local_rot is None else L.require_cuda_f32("g_local_rot", g_local_rot, (J, 4))
        ggt = None if g_global_trans is None else L.require_cuda_f32("g_global_trans", g_global_trans.reshape(-1), (3,))
        g = L.FrameGrads()
        g.dL_dcolor, g.dL_ddepth, g.dL_dalpha, g.raster_workspace = gc.data_ptr(), L.ptr(gd), L.ptr(ga), ws.data_ptr()
        g.dL_dxyz, g.dL_dmeans2D, g.dL_dfeatures_dc, g.dL_dfeatures_rest = g_xyz.data_ptr(), g_m2d.data_ptr(), g_dc.data_ptr(), g_rest.data_ptr()
        g.dL_dopacity, g.dL_dscaling, g.dL_drotation, g.dL_dd_scaling = g_op.data_ptr(), g_sc.data_ptr(), g_rot.data_ptr(), None
        g.dL_dtransforms, g.dL_dnode_radius_log, g.dL_dglobal_trans_skinning = dG.data_ptr(), drho.data_ptr(), dgt_s.data_ptr()
        g.dL_dmotion_mask, g.dL_dweight_mod, g.lbs_workspace = L.ptr(dmask), None, lws.data_ptr()
        g.dL_dd_nodes, g.g_local_rot, g.g_global_trans = L.ptr(gn), L.ptr(gq), L.ptr(ggt)
        g.dL_dlocal_rot, g.dL_dglobal_trans, g.pose_workspace, g.pose_flat_grads = dq.data_ptr(), dgt.data_ptr(), dzs.data_ptr(), flat.data_ptr()
        fr.cfg.sparse_zero = 0
        try:
            L.check(lib.riggs_frame_backward(C.byref(fr), C.byref(g), L.stream_ptr()), "riggs_frame_backward")
        except Exception:
            from .rasterizer import _WORKSPACES
            _WORKSPACES.clear()
            raise
        _LAST_WORKSPACE[:] = [ws, N]
        # (one split + a view per matrix: a slice and a view per parameter were 30 us of an eager frame)
        grads = [g_ if p.dim() == 1 else g_.view(p.shape) for g_, p in zip(flat.split_with_sizes([p.numel() for p in params]), params)]
        gmask = dmask.reshape(mask_shape) if need_mask else None
        return (None, drho, gmask, g_xyz, g_m2d, g_dc, g_rest, g_op, g_sc, g_rot, None, *grads)


def _covered(pc, sw, pipe, t, kw):
    if kw or pipe.compute_cov3D_python or pipe.convert_SHs_python or not pc._xyz.is_cuda:
        return False
    pn = sw.pose_net
    key = (sw.use_skinning_weight_mlp, sw.use_template_offsets, sw.K, pn.rotation_predictor.out_features, sw.nodes.shape[0], t.device)
    cached = getattr(sw, "_frame_entry_ok", None)
    if cached is None or cached[0] != key:  # (the network's shape does not change under training: checked once per configuration)
        ok = (not sw.use_skinning_weight_mlp and not sw.use_template_offsets and sw.K <= 0 and pn._fusable(t[0])
              and pn.rotation_predictor.out_features == 4 * sw.nodes.shape[0])
        cached = sw._frame_entry_ok = (key, ok)
    return cached[1]


def deform_render(viewpoint_camera, pc, sw, pipe, bg_color, scaling_modifier=1.0, arena: RasterArena = None, **render_kwargs):
    """``d = sw(pc.get_xyz.detach(), sw.expand_time(cam.fid), motion_mask=pc.motion_mask)`` followed by
    ``render(cam, pc, pipe, bg, d["d_xyz"], d["d_rotation"], d["d_scaling"], ...)`` — as one autograd node where the frame
    entry covers the configuration (see the module docstring), as exactly those two calls otherwise.  ``arena``: the persistent
    instance arena (riggs_amd.rasterizer.RasterArena); one per Gaussian model is kept here when none is given."""
    sw = getattr(sw, "deform", sw)  # (a SkeletonModel wrapper)
    if arena is None:
        # (kept ON the model — it is that model's frame-to-frame state and dies with it; a cache keyed by id(pc) would hand a
        # new model that got a dead one's address an arena with a stale instance count)
        arena = getattr(pc, "_frame_arena", None)
        if arena is None or getattr(pc, "_frame_arena_n", None) != pc._xyz.shape[0]:
            arena = pc._frame_arena = RasterArena()
            pc._frame_arena_n = pc._xyz.shape[0]  # (densification changed N: the count of the last frame says nothing any more)
    t = sw.expand_time(viewpoint_camera.fid)
    if not _covered(pc, sw, pipe, t, render_kwargs) or arena.last_R < 0:
        dv = sw(pc.get_xyz.detach(), t, motion_mask=pc.motion_mask)
        pkg = render(viewpoint_camera, pc, pipe, bg_color, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"],
                     scaling_modifier=scaling_modifier, fused=True, arena=arena, **render_kwargs)
        for k in ("d_nodes", "local_rotation", "global_trans", "d_xyz", "d_rotation", "d_scaling"):
            dict.__setitem__(pkg, k, dv[k])
        return pkg
    pn = sw.pose_net
    params = []
    for layer in pn.net._modules.values():  # (straight from the registries: Module.__getattr__ per access is 10 us a frame)
        pd = layer._parameters
        params += [pd["weight"], pd["bias"]]
    for head in (pn._modules["rotation_predictor"], pn._modules["translation_predictor"]):
        params += [head._parameters["weight"], head._parameters["bias"]]
    settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=pipe.debug)
    iso = bool(getattr(pc, "use_isotropic_gs", False))
    scaling = pc._scaling[..., :1] if iso else pc._scaling
    mask = pc.motion_mask
    if mask is not None and not isinstance(mask, torch.Tensor):
        mask = None if float(mask) == 1.0 else torch.full((pc._xyz.shape[0], 1), float(mask), device=pc._xyz.device)
    pts = _zero_points(pc._xyz)
    color, radii, depth, alpha, d_nodes, local_rot, global_trans, d_xyz, d_rot = _FrameFn.apply(
        t[0].reshape(1), sw._node_radius, mask, pc._xyz, pts, pc._features_dc, pc._features_rest, pc._opacity, scaling,
        pc._rotation, (sw, settings, arena, iso), *params)
    zs = getattr(sw, "_zero_scaling", None)
    if zs is None or zs.shape[0] != d_xyz.shape[0] or zs.device != d_xyz.device:
        zs = sw._zero_scaling = torch.zeros(d_xyz.shape[0], 3, device=d_xyz.device)
    return RenderPkg({"render": color, "viewspace_points": pts, "visibility_filter": None, "radii": radii, "depth": depth,
                      "alpha": alpha, "bg_color": bg_color, "d_nodes": d_nodes, "local_rotation": local_rot,
                      "global_trans": global_trans, "d_xyz": d_xyz, "d_rotation": d_rot, "d_scaling": zs})
