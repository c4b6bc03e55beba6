"""Image loss of the trainer with the reference's names and signatures (/root/reference/utils/loss_utils.py:17-18 ``l1_loss``,
:47-77 ``ssim``), computed by two HIP launches (csrc/loss.hip) instead of ~25 torch ops and their autograd replay.

The trainer calls ``l1_loss(image, gt)`` and ``ssim(image, gt)`` back to back on the same pair
(/root/reference/train_rig.py:508-509); both come out of ONE fused forward here — the second call finds the first one's
autograd node (matched by tensor identity and version) — and ``loss.backward()`` runs ONE fused backward that produces
``dL/dimage`` for the rasterizer.  No CPU / eager fallback.
"""
from __future__ import annotations

import weakref

import torch

from . import _lib as L


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim=0.2):
        C, H, W = image.shape
        lib = L.lib()
        state = torch.empty(lib.riggs_l1_ssim_state_floats(C, H, W), dtype=torch.float32, device=image.device)
        out3 = torch.empty(3, dtype=torch.float32, device=image.device)
        L.check(lib.riggs_l1_ssim_forward(C, H, W, image.data_ptr(), gt.data_ptr(), float(lambda_dssim), state.data_ptr(),
                                          out3.data_ptr(), L.stream_ptr()), "riggs_l1_ssim_forward")
        ctx.save_for_backward(image, gt, state)
        ctx.lam = float(lambda_dssim)
        ctx.set_materialize_grads(False)
        return out3[0], out3[1], out3[2]

    @staticmethod
    def backward(ctx, g_l1, g_ssim, g_loss=None):
        global _last
        _last = None  # this node is consumed: a later call on the same tensors must build a new one
        image, gt, state = ctx.saved_tensors
        C, H, W = image.shape
        dx = torch.empty_like(image)
        f = lambda g: None if g is None else g.to(torch.float32).contiguous()  # noqa: E731
        g_l1, g_ssim, g_loss = f(g_l1), f(g_ssim), f(g_loss)
        L.check(L.lib().riggs_l1_ssim_backward(C, H, W, image.data_ptr(), gt.data_ptr(), state.data_ptr(), ctx.lam,
                                               L.ptr(g_l1), L.ptr(g_ssim), L.ptr(g_loss), dx.data_ptr(), L.stream_ptr()),
                "riggs_l1_ssim_backward")
        return dx, None, None


_last = None  # (weakref(image), version, weakref(gt), version, (l1, ssim))


def _chw(t, name):
    if t.dim() == 4 and t.shape[0] == 1:
        t = t[0]
    if t.dim() != 3:
        raise NotImplementedError("%s must be (C, H, W) or (1, C, H, W): the trainer's case (train_rig.py:508-509)" % name)
    return L.require_cuda_f32(name, t)


def l1_ssim(image, gt):
    """Both scalars of the image loss from one fused forward: ``(mean |image - gt|, ssim(image, gt))``."""
    global _last
    if _last is not None:
        wi, vi, wg, vg, out = _last
        if wi() is image and wg() is gt and image._version == vi and gt._version == vg and torch.is_grad_enabled() == out[0].requires_grad:
            return out
    out = _L1SSIM.apply(_chw(image, "image"), _chw(gt, "gt").detach())[:2]
    _last = (weakref.ref(image), image._version, weakref.ref(gt), gt._version, out)
    return out


def image_loss(image, gt, lambda_dssim=0.2):
    """The trainer's ``loss_img = (1 - lambda) * l1_loss + lambda * (1 - ssim)`` (train_rig.py:508-509) with the combination
    done inside the fused kernels (no scalar glue launches): returns ``(loss_img, Ll1)``."""
    l1, _, loss = _L1SSIM.apply(_chw(image, "image"), _chw(gt, "gt").detach(), float(lambda_dssim))
    return loss, l1


def l1_loss(network_output, gt):
    return l1_ssim(network_output, gt)[0]


def ssim(img1, img2, window_size=11, size_average=True):
    if window_size != 11 or not size_average:
        raise NotImplementedError("the HIP kernel implements the trainer's call: window_size=11, size_average=True")
    return l1_ssim(img1, img2)[1]


# ---- skeleton projection loss (train_rig.py:309-314) ---------------------------------------------------------------------
class _SkeletonProjection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d_nodes, parents, t, view, thinned, weight, count, fx, fy, cx, cy):
        J, S, M = d_nodes.shape[0], t.shape[0], thinned.shape[0]
        lib = L.lib()
        n_state = lib.riggs_skeleton_projection_state_floats(J, S, M)
        state = torch.empty(max(int(n_state), 2) // 2 + 1, dtype=torch.float64, device=d_nodes.device)  # 8-byte aligned
        loss2 = torch.empty(2, dtype=torch.float32, device=d_nodes.device)
        L.check(lib.riggs_skeleton_projection_forward(J, S, M, parents.data_ptr(), d_nodes.data_ptr(), t.data_ptr(),
                                                      view.data_ptr(), fx, fy, cx, cy, thinned.data_ptr(), L.ptr(count),
                                                      L.ptr(weight), state.data_ptr(), loss2.data_ptr(), L.stream_ptr()),
                "riggs_skeleton_projection_forward")
        ctx.save_for_backward(d_nodes, parents, t, view, thinned, state, weight, count)
        ctx.intr = (fx, fy, cx, cy)
        ctx.set_materialize_grads(False)
        return loss2[0], loss2[1]

    @staticmethod
    def backward(ctx, g_loss, g_weighted):
        d_nodes, parents, t, view, thinned, state, weight, count = ctx.saved_tensors
        J, S, M = d_nodes.shape[0], t.shape[0], thinned.shape[0]
        if g_loss is None and g_weighted is None:
            return (None,) * 11
        f = lambda g: None if g is None else g.to(torch.float32).contiguous()  # noqa: E731
        g_loss, g_weighted = f(g_loss), f(g_weighted)
        grad = torch.empty_like(d_nodes)
        fx, fy, cx, cy = ctx.intr
        L.check(L.lib().riggs_skeleton_projection_backward(J, S, M, parents.data_ptr(), d_nodes.data_ptr(), t.data_ptr(),
                                                           view.data_ptr(), fx, fy, cx, cy, thinned.data_ptr(), L.ptr(count),
                                                           L.ptr(weight), state.data_ptr(), L.ptr(g_loss), L.ptr(g_weighted),
                                                           grad.data_ptr(), L.stream_ptr()),
                "riggs_skeleton_projection_backward")
        return (grad,) + (None,) * 10


def sampling_steps(joints, parents, num_sample=512):
    """Line parameters of ``TrainRig.sampling_skeleton_points`` (/root/reference/train_rig.py:264-272): ``linspace(0, 1, S)``
    with ``S = int(max bone length / (sum of bone lengths / num_sample))`` evaluated in float32 exactly as the reference does
    (one host sync for ``int()``, as there).  FK is rigid, so S only changes when the rest joints do: a captured training
    iteration computes it once and passes it to ``cal_skeleton_loss(..., t=...)``."""
    j = joints.detach()
    distance = (j[1:] - j[parents[1:].long()]).norm(dim=-1)
    each_distance = distance.sum() / num_sample
    return torch.linspace(0, 1, int(distance.max() / each_distance), device=j.device)


def camera_intrinsics(viewpoint_cam):
    """fx, fy, cx, cy of ``project_nodes_to_2d_elements`` (/root/reference/utils/other_utils.py:101-117)."""
    import math
    H, W = int(viewpoint_cam.image_height), int(viewpoint_cam.image_width)
    fy = H / (2 * math.tan(viewpoint_cam.FoVy * 0.5))
    fx = W / (2 * math.tan(viewpoint_cam.FoVx * 0.5))
    K = getattr(viewpoint_cam, "K", None)
    if K is not None:
        return float(fx), float(fy), float(K[0][2]), float(K[1][2])
    return float(fx), float(fy), W / 2, H / 2


def cal_skeleton_loss(d_nodes, parents, viewpoint_cam, t=None, num_sample=512, weight=None, pixel_count=None):
    """``TrainRig.cal_skeleton_loss(d_nodes, viewpoint_cam)`` (/root/reference/train_rig.py:309-314) with the skeleton's
    ``parents`` passed explicitly: points sampled on the posed bones, projected with the camera (elements are (row, col)) and
    compared with ``viewpoint_cam.thinned`` by the two-sided L1 chamfer distance; differentiable w.r.t. ``d_nodes``.

    With ``weight`` (a device scalar: the trainer's robust per-frame weight, train_rig.py:465-467) the pair
    ``(loss, weight * loss)`` is returned, the product formed inside the kernels instead of two more launches.
    ``pixel_count`` (a device int32 scalar, 1 <= count <= len(thinned)) marks the valid rows of ``thinned``, which is then a
    buffer of fixed capacity: a captured graph serves frames of any pixel count."""
    d_nodes = L.require_cuda_f32("d_nodes", d_nodes).contiguous()
    if d_nodes.dim() != 2 or d_nodes.shape[1] != 3 or d_nodes.shape[0] < 2:
        raise L.RiggsHipError("d_nodes must be (J >= 2, 3)")
    par = parents.to(device=d_nodes.device, dtype=torch.int32).contiguous()
    if t is None:
        t = sampling_steps(d_nodes, par, num_sample)
    t = L.require_cuda_f32("t", t).contiguous()
    thinned = L.require_cuda_f32("viewpoint_cam.thinned", viewpoint_cam.thinned).contiguous()
    if thinned.dim() != 2 or thinned.shape[1] != 2:
        raise L.RiggsHipError("viewpoint_cam.thinned must be (M, 2) (row, col)")
    if t.shape[0] == 0 or thinned.shape[0] == 0:
        raise L.RiggsHipError("empty point set: the chamfer distance of the reference is undefined")
    view = L.require_cuda_f32("viewpoint_cam.world_view_transform", viewpoint_cam.world_view_transform, (4, 4)).contiguous()
    fx, fy, cx, cy = camera_intrinsics(viewpoint_cam)
    if weight is not None:
        weight = L.require_cuda_f32("weight", weight).reshape(1)
    if pixel_count is not None:
        if not pixel_count.is_cuda or pixel_count.dtype != torch.int32:
            raise L.RiggsHipError("pixel_count must be a CUDA(HIP) int32 scalar tensor")
        pixel_count = pixel_count.reshape(1)
    loss, weighted = _SkeletonProjection.apply(d_nodes, par, t, view, thinned, weight, pixel_count, fx, fy, cx, cy)
    return loss if weight is None else (loss, weighted)


class ProjectionLossWeights:
    """The robust per-frame weight the trainer puts on the skeleton projection loss (/root/reference/train_rig.py:462-467):
    the latest loss of every training frame is remembered (initialised to 1e5, :114), sigma = median / 2,
    weight = lambda * exp(-l^2 / (2 sigma^2))."""

    def __init__(self, num_frames, lambda_deformed_node_prjection=1e-3, init=1.0e5):
        self.all_nodes_projection_loss = torch.full((num_frames,), float(init))
        self.lam = float(lambda_deformed_node_prjection)

    def update(self, uid, loss_value):
        self.all_nodes_projection_loss[uid] = float(loss_value)
        sigma = self.all_nodes_projection_loss.median() / 2.0
        return self.lam * torch.exp(-self.all_nodes_projection_loss[uid] ** 2 / (2.0 * sigma ** 2))
