"""The Gaussian optimizer of the reference — ``torch.optim.Adam(l, lr=0.0, eps=1e-15)`` built by
``GaussianModel.training_setup`` (/root/reference/scene/gaussian_model.py:197-221) and stepped at
/root/reference/train_rig.py:527 — with ``step()`` as ONE HIP launch over all parameter groups
(``riggs_adam_step``, csrc/optim.hip) instead of ~10 elementwise passes per tensor.

``FusedAdam`` *is* a ``torch.optim.Adam``: same constructor, ``param_groups``, per-parameter ``state`` with the keys
``step`` / ``exp_avg`` / ``exp_avg_sq``, ``state_dict`` — so the reference's optimizer surgery (``replace_tensor_to_optimizer``,
``_prune_optimizer``, ``cat_tensors_to_optimizer``: scene/gaussian_model.py:338-420) and ``update_learning_rate`` work on it
unchanged.  Only the update itself is replaced; there is no CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

_MAX = 8  # tensors per launch (ADAM_MAX_GROUPS in csrc/optim.hip)


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if weight_decay != 0 or amsgrad or kw.get("maximize", False):
            raise NotImplementedError("FusedAdam implements the reference's configuration: plain Adam "
                                      "(weight_decay=0, amsgrad=False, maximize=False)")
        kw.pop("foreach", None), kw.pop("fused", None), kw.pop("capturable", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.lib()
        by_cfg = {}
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("FusedAdam: plain Adam only")
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                L.require_cuda_f32("parameter", p)
                if not p.is_contiguous():
                    raise L.RiggsHipError("FusedAdam needs contiguous parameters")
                st = self.state[p]
                if len(st) == 0:  # same lazy initialisation as torch.optim.Adam
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    st["exp_avg"], st["exp_avg_sq"] = m, v = m.contiguous(), v.contiguous()
                g = L.require_cuda_f32("gradient", p.grad, tuple(p.shape))
                key = (group["betas"][0], group["betas"][1], group["eps"])
                by_cfg.setdefault(key, []).append((p, g, m, v, float(group["lr"]), int(st["step"].item())))
        st_ptr = L.stream_ptr()
        for (b1, b2, eps), items in by_cfg.items():
            for i in range(0, len(items), _MAX):
                chunk = items[i:i + _MAX]
                n = len(chunk)
                arr = lambda k: (C.c_void_p * n)(*[t[k].data_ptr() for t in chunk])  # noqa: E731
                numel = (C.c_int64 * n)(*[t[0].numel() for t in chunk])
                lr = (C.c_double * n)(*[t[4] for t in chunk])
                steps = (C.c_int64 * n)(*[t[5] for t in chunk])
                L.check(lib.riggs_adam_step(n, arr(0), arr(1), arr(2), arr(3), numel, lr, steps, float(b1), float(b2),
                                            float(eps), st_ptr), "riggs_adam_step")
        return loss


def densify_stats(viewspace_grad, update_filter, xyz_gradient_accum, denom, radii=None, max_radii2D=None):
    """In place: ``xyz_gradient_accum[f] += ||viewspace_grad[f, :2]||``, ``denom[f] += 1`` (add_densification_stats,
    scene/gaussian_model.py:516-518) and optionally ``max_radii2D[f] = max(max_radii2D[f], radii[f])``
    (train_rig.py:333-335) in one HIP launch."""
    N = viewspace_grad.shape[0]
    vg = L.require_cuda_f32("viewspace gradient", viewspace_grad, (N, 3))
    f = update_filter
    if f.dtype != torch.bool or not f.is_cuda or f.numel() != N:
        raise L.RiggsHipError("update_filter must be a CUDA bool tensor with one entry per Gaussian")
    f = f.reshape(-1).contiguous()
    acc = L.require_cuda_f32("xyz_gradient_accum", xyz_gradient_accum)
    den = L.require_cuda_f32("denom", denom)
    if acc.numel() != N or den.numel() != N or acc.data_ptr() != xyz_gradient_accum.data_ptr() or den.data_ptr() != denom.data_ptr():
        raise L.RiggsHipError("xyz_gradient_accum / denom must be contiguous with one entry per Gaussian")
    rp = mp = None
    if max_radii2D is not None:
        if radii is None or radii.dtype != torch.int32 or not radii.is_contiguous():
            raise L.RiggsHipError("radii must be the contiguous int32 tensor the rasterizer returned")
        mr = L.require_cuda_f32("max_radii2D", max_radii2D, (N,))
        if mr.data_ptr() != max_radii2D.data_ptr():
            raise L.RiggsHipError("max_radii2D must be contiguous")
        rp, mp = radii.data_ptr(), mr.data_ptr()
    L.check(L.lib().riggs_densify_stats(N, vg.data_ptr(), f.data_ptr(), rp, acc.data_ptr(), den.data_ptr(), mp,
                                        L.stream_ptr()), "riggs_densify_stats")
