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
