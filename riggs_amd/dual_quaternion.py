"""Dual-quaternion blending — host mirror of /root/reference/utils/dual_quaternion.py (same function names, argument meaning
and results): ``DQBlending`` :168-179, ``interpolate`` :182-187 and ``transformation_blending`` :190-197 run on the HIP
kernels of csrc/dq.hip (C ABI: riggs_dqb_forward / riggs_dqb_backward, include/riggs_hip.h) as ONE autograd node each —
the per-row work (N rows x K transforms: QT2DQ, the weighted sum, DQ2QT, matrix_to_quaternion) never materialises an
(N, K, 8) tensor.  ``QT2DQ`` / ``DQ2QT`` and the quaternion helpers are kept as small torch compositions for callers that
use them on node-sized tensors.  The reference never calls this module on the skeleton path, so nothing in
``SkeletonWarp`` depends on it; ``dqb_skinning`` below is the composition a caller who wants dual-quaternion skinning with
the skeleton's weights uses.

Two properties of the reference are reproduced on purpose (see oracle/dq_ref.py): ``torch.nn.functional.normalize(q)`` in
QT2DQ acts on dim=1 — the quaternion axis of a 2-D ``q``, the NODE axis of a 3-D one —, and the dual part is
``standardize_quaternion((0, t) * q) / 2`` while the real part keeps q's sign.  No CPU fallback: the HIP entry points
reject CPU tensors.
"""
from __future__ import annotations

import torch

from . import _lib as L


# ---- node-sized helpers (torch ops on device tensors; dual_quaternion.py:15-132) ----------------------------------------
def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """(..., 3, 3) -> (..., 4) real part first; the candidate with the largest denominator, no sign standardisation (:15-74)."""
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    m = matrix.reshape(matrix.shape[:-2] + (9,))
    d0, d1, d2 = m[..., 0], m[..., 4], m[..., 8]
    s = torch.stack([1.0 + d0 + d1 + d2, 1.0 + d0 - d1 - d2, 1.0 - d0 + d1 - d2, 1.0 - d0 - d1 + d2], -1)
    q_abs = torch.where(s > 0, torch.sqrt(s.clamp_min(1e-38)), torch.zeros_like(s))
    a, b, c = m[..., 7] - m[..., 5], m[..., 2] - m[..., 6], m[..., 3] - m[..., 1]
    p, q, r = m[..., 3] + m[..., 1], m[..., 2] + m[..., 6], m[..., 5] + m[..., 7]
    sq = q_abs * q_abs
    cand = torch.stack([torch.stack([sq[..., 0], a, b, c], -1), torch.stack([a, sq[..., 1], p, q], -1),
                        torch.stack([b, p, sq[..., 2], r], -1), torch.stack([c, q, r, sq[..., 3]], -1)], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    best = q_abs.argmax(-1)
    return torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4)))[..., 0, :]


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    """(..., 4) -> (..., 3, 3) with the 2 / |q|^2 factor (:77-94)."""
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def standardize_quaternion(quaternions: torch.Tensor) -> torch.Tensor:
    return torch.where(quaternions[..., 0:1] < 0, -quaternions, quaternions)


def quaternion_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return standardize_quaternion(quaternion_raw_multiply(a, b))


def dualquaternion_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    ar, br, ai, bi = a[..., :4], b[..., :4], a[..., 4:], b[..., 4:]
    return torch.cat([quaternion_multiply(ar, br), quaternion_multiply(ai, br) + quaternion_multiply(ar, bi)], -1)


def conjugation(q):
    if q.shape[-1] == 4:
        return torch.cat([q[..., :1], -q[..., 1:]], -1)
    if q.shape[-1] == 8:
        return torch.cat([q[..., :1], -q[..., 1:4], q[..., 4:5], -q[..., 5:]], -1)
    raise TypeError(f"q should be of [..., 4] or [..., 8] but got {q.shape}!")


def QT2DQ(q, t, rot_as_q=True):
    """(q, t) -> dual quaternion (..., 8) (:135-143); ``normalize`` acts on dim=1 exactly as in the reference."""
    if not rot_as_q:
        q = matrix_to_quaternion(q)
    q = torch.nn.functional.normalize(q)
    t4 = torch.cat([torch.zeros_like(t[..., :1]), t], -1)
    return torch.cat([q, quaternion_multiply(t4, q) / 2], -1)


def DQ2QT(dq, rot_as_q=False):
    """dual quaternion (..., 8) -> (R | q, t) (:146-165)."""
    real, imag = dq[..., :4], dq[..., 4:]
    n = real.norm(dim=-1, keepdim=True).clamp(min=1e-8)
    real, imag = real / n, imag / n
    w0, x0, y0, z0 = torch.unbind(real, -1)
    w1, x1, y1, z1 = torch.unbind(imag, -1)
    t = 2 * torch.stack([-w1 * x0 + x1 * w0 - y1 * z0 + z1 * y0, -w1 * y0 + x1 * z0 + y1 * w0 - z1 * x0,
                         -w1 * z0 - x1 * y0 + y1 * x0 + z1 * w0], -1)
    R = torch.stack([1 - 2 * y0 ** 2 - 2 * z0 ** 2, 2 * x0 * y0 - 2 * w0 * z0, 2 * x0 * z0 + 2 * w0 * y0,
                     2 * x0 * y0 + 2 * w0 * z0, 1 - 2 * x0 ** 2 - 2 * z0 ** 2, 2 * y0 * z0 - 2 * w0 * x0,
                     2 * x0 * z0 - 2 * w0 * y0, 2 * y0 * z0 + 2 * w0 * x0, 1 - 2 * x0 ** 2 - 2 * y0 ** 2], -1).reshape(w0.shape + (3, 3))
    return (matrix_to_quaternion(R), t) if rot_as_q else (R, t)


# ---- the per-row blend: HIP ----------------------------------------------------------------------------------------------
class _DQBlend(torch.autograd.Function):
    """riggs_dqb_forward / riggs_dqb_backward.  q (K, 4) [shared] or (N, K, 4) [rows], t alike, weights (N, K)."""

    @staticmethod
    def forward(ctx, q, t, weights, shared, norm_over_nodes, out_mode):
        ctx.set_materialize_grads(False)
        N, K = weights.shape
        q = L.require_cuda_f32("q", q, (K, 4) if shared else (N, K, 4))
        t = L.require_cuda_f32("t", t, (K, 3) if shared else (N, K, 3))
        weights = L.require_cuda_f32("weights", weights, (N, K))
        f32 = dict(dtype=torch.float32, device=weights.device)
        rot = torch.empty((N, (9, 4, 16)[out_mode]), **f32)
        tr = torch.empty((N, 3), **f32) if out_mode != 2 else None
        L.check(L.lib().riggs_dqb_forward(N, K, int(shared), int(norm_over_nodes), out_mode, q.data_ptr(), t.data_ptr(),
                                          weights.data_ptr(), rot.data_ptr(), L.ptr(tr), L.stream_ptr()), "riggs_dqb_forward")
        ctx.save_for_backward(q, t, weights)
        ctx.cfg = (bool(shared), bool(norm_over_nodes), out_mode)
        if out_mode == 2:
            return rot.view(N, 4, 4)
        return (rot.view(N, 3, 3) if out_mode == 0 else rot), tr

    @staticmethod
    def backward(ctx, g_rot, g_t=None):
        q, t, weights = ctx.saved_tensors
        shared, norm_over_nodes, out_mode = ctx.cfg
        N, K = weights.shape
        f32 = dict(dtype=torch.float32, device=weights.device)
        if g_rot is None:
            g_rot = torch.zeros((N, (9, 4, 16)[out_mode]), **f32)
        g_rot = L.require_cuda_f32("g_rot", g_rot.reshape(N, -1), (N, (9, 4, 16)[out_mode]))
        g_t = None if (g_t is None or out_mode == 2) else L.require_cuda_f32("g_t", g_t, (N, 3))
        gq, gt = torch.empty_like(q), torch.empty_like(t)
        gw = torch.empty_like(weights) if ctx.needs_input_grad[2] else None
        lib = L.lib()
        ws = torch.empty(max(1, int(lib.riggs_dqb_backward_workspace_floats(N, K, int(shared)))), **f32)
        L.check(lib.riggs_dqb_backward(N, K, int(shared), int(norm_over_nodes), out_mode, q.data_ptr(), t.data_ptr(),
                                       weights.data_ptr(), g_rot.data_ptr(), L.ptr(g_t), gq.data_ptr(), gt.data_ptr(), L.ptr(gw),
                                       ws.data_ptr(), L.stream_ptr()), "riggs_dqb_backward")
        return gq, gt, gw, None, None, None


def _blend_general(q, t, weights, out_mode):
    """Shapes the kernels have no form for — leading batch dimensions beyond one, more than 8 transforms of its own per row —
    through the composition the reference itself is (:168-179): QT2DQ -> weighted sum over the node axis -> DQ2QT, on the
    tensors' device with this module's torch helpers.  Not a CPU fallback: device tensors stay on the device."""
    dq = QT2DQ(q, t)
    return DQ2QT((dq * weights[..., None]).sum(dim=-2), rot_as_q=(out_mode == 1))


def _blend(q, t, weights, out_mode):
    """Shape dispatch of DQBlending: which axis QT2DQ's ``normalize`` hits follows from q's rank, as in the reference."""
    if weights.dim() != 2 or not weights.is_cuda:
        if not weights.is_cuda:
            raise L.RiggsHipError("DQBlending: CUDA(HIP) tensors required — the product path is GPU-only")
        return _blend_general(q, t, weights, out_mode)
    N, K = weights.shape
    if q.dim() == 2:                       # (K, 4): shared nodes, per-quaternion normalisation
        return _DQBlend.apply(q, t, weights, True, False, out_mode)
    if q.dim() == 3 and q.shape[0] == 1:   # (1, K, 4): shared nodes, F.normalize on the node axis
        return _DQBlend.apply(q[0], t.reshape(K, 3), weights, True, True, out_mode)
    if q.dim() == 3 and q.shape[0] == N:   # (N, K, 4): every row its own K transforms
        if K > 8:  # (the rows kernel keeps a row's K transforms in registers: 8 at most)
            return _blend_general(q, t, weights, out_mode)
        return _DQBlend.apply(q, t, weights, False, True, out_mode)
    return _blend_general(q, t, weights, out_mode)


def DQBlending(q, t, weights, rot_as_q=True):
    """q (..., k, 4), t (..., k, 3), weights (..., k) -> (q_ (..., 4) | R (..., 3, 3), t_ (..., 3)) (:168-179)."""
    return _blend(q, t, weights, 1 if rot_as_q else 0)


def interpolate(q0, t0, q1, t1, weight, rot_as_q=True):
    """dq0 * weight + dq1 * (1 - weight) back to (q | R, t) (:182-187); q0 / q1 (M, 4), weight a scalar or (M, 1)."""
    if q0.dim() != 2:
        raise NotImplementedError("interpolate: (M, 4) quaternions")
    M = q0.shape[0]
    w = weight if isinstance(weight, torch.Tensor) else torch.full((1, 1), float(weight), device=q0.device)
    w = w.reshape(-1, 1).expand(M, 1).to(torch.float32)
    return _DQBlend.apply(torch.stack([q0, q1], 1), torch.stack([t0, t1], 1), torch.cat([w, 1 - w], 1), False, False,
                          1 if rot_as_q else 0)


def transformation_blending(transformations, weights):
    """(K, 4, 4) rigid transforms, weights (N, K) -> (N, 4, 4) (:190-197).  The reference goes R -> q (matrix_to_quaternion)
    -> DQBlending(q[None], ...) [node-axis normalisation] -> quaternion_to_matrix; the kernel writes [R | t; 0 0 0 1] of the
    blended unit dual quaternion directly: the same matrix (quaternion_to_matrix(matrix_to_quaternion(R)) = R for a rotation)
    and the same gradients (the detour's extra Jacobian acts along the quaternion's norm, which DQ2QT's normalisation
    projects out)."""
    Rs, Ts = transformations[:, :3, :3], transformations[:, :3, 3]
    return _DQBlend.apply(matrix_to_quaternion(Rs), Ts, weights, True, True, 2)


def dqb_skinning(x, transforms, weights):
    """Dual-quaternion skinning of points: ``x`` (N, 3), ``transforms`` (K, 3, 4) or (K, 4, 4) rigid node transforms (e.g. the
    skeleton's bone transforms ``G[1:]`` from forward kinematics), ``weights`` (N, K) (e.g. ``deform_by_pose(...)["nn_weight"]``)
    -> (x' (N, 3), R (N, 3, 3), t (N, 3)) with x' = R x + t, R / t = the blended transform of transformation_blending."""
    T = transformation_blending(transforms if transforms.shape[-2] == 4 else
                                torch.cat([transforms, transforms.new_tensor([0.0, 0.0, 0.0, 1.0]).expand(transforms.shape[0], 1, 4)], 1),
                                weights)
    R, t = T[:, :3, :3], T[:, :3, 3]
    return torch.einsum("nij,nj->ni", R, x) + t, R, t
