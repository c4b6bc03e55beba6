"""ORACLE (test infrastructure, NOT product code).

Dual-quaternion blending of rigid transforms: a numpy restatement (forward AND a hand-derived backward, any float dtype —
the tests run it in float64) of /root/reference/utils/dual_quaternion.py:
  QT2DQ :135-143, DQ2QT :146-165, DQBlending :168-179, interpolate :182-187, transformation_blending :190-197, with
  quaternion_raw_multiply / quaternion_multiply / standardize_quaternion :97-113 and matrix_to_quaternion :15-74.
Only ``tests/`` may import this module; the product path (riggs_amd/dual_quaternion.py + csrc/dq.hip) never does.

Parity status: PINNED by tests/golden/dqb_*.npz = outputs and autograd gradients of the reference's OWN functions run on the
CPU (tests/golden/make_golden.py: fixture_dqb), checked by tests/test_oracle_dq.py.

Two properties of the reference that a "textbook" DQB would not have, both reproduced here (results parity):
  * QT2DQ normalises with ``torch.nn.functional.normalize(q)``, whose default axis is dim=1.  For a 2-D ``q`` (K, 4) that
    is the quaternion axis; for a 3-D ``q`` (B, K, 4) — what DQBlending's docstring describes and what
    transformation_blending passes (``qs[None]``) — it is the NODE axis: every quaternion COMPONENT is divided by its norm
    over the K nodes.  ``norm_over_nodes`` selects between the two.
  * the dual part is ``quaternion_multiply((0, t), q) / 2`` and quaternion_multiply STANDARDISES its product to a
    non-negative real part (:110-113) while the real part of the dual quaternion keeps q's sign: a node whose product
    (0, t) * q has a negative real part enters the blend with its dual part negated.
Quaternions are (w, x, y, z).
"""
from __future__ import annotations

import numpy as np


def _raw_mul(a, b):
    """quaternion_raw_multiply (:97-104)."""
    aw, ax, ay, az = (a[..., i] for i in range(4))
    bw, bx, by, bz = (b[..., i] for i in range(4))
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], -1)


def qt2dq(q, t, norm_over_nodes):
    """QT2DQ (:135-143) for q (B, K, 4), t (B, K, 3).  Returns (dq (B, K, 8), cache)."""
    if norm_over_nodes:   # F.normalize(q) on a 3-D tensor: dim=1 is the node axis; eps 1e-12
        nrm = np.maximum(np.sqrt((q * q).sum(-2, keepdims=True)), 1e-12)   # (B, 1, 4)
    else:                 # 2-D q: the quaternion axis
        nrm = np.maximum(np.sqrt((q * q).sum(-1, keepdims=True)), 1e-12)   # (B, K, 1)
    qn = q / nrm
    tq = np.concatenate([np.zeros_like(t[..., :1]), t], -1)
    p = _raw_mul(tq, qn)
    sgn = np.where(p[..., :1] < 0, -1.0, 1.0).astype(q.dtype)              # standardize_quaternion (:93-94)
    dq = np.concatenate([qn, sgn * p * 0.5], -1)
    return dq, (q, t, qn, nrm, sgn)


def qt2dq_backward(cache, g_dq, norm_over_nodes):
    q, t, qn, nrm, sgn = cache
    g_real, g_img = g_dq[..., :4], g_dq[..., 4:]
    gp = g_img * sgn * 0.5                        # p = (0, t) * qn ; image = sgn * p / 2 (sgn piecewise constant)
    ax, ay, az = t[..., 0], t[..., 1], t[..., 2]
    bw, bx, by, bz = (qn[..., i] for i in range(4))
    gw, gx, gy, gz = (gp[..., i] for i in range(4))
    g_qn = g_real + np.stack([ax * gx + ay * gy + az * gz,
                              -ax * gw + az * gy - ay * gz,
                              -ay * gw - az * gx + ax * gz,
                              -az * gw + ay * gx - ax * gy], -1)
    g_t = np.stack([-bx * gw + bw * gx - bz * gy + by * gz,
                    -by * gw + bz * gx + bw * gy - bx * gz,
                    -bz * gw - by * gx + bx * gy + bw * gz], -1)
    axis = -2 if norm_over_nodes else -1
    # y = q / max(|q|, eps) along `axis`:  g_q = (g_y - y (y . g_y)) / |q|   (the clamp is inactive for any real input)
    g_q = (g_qn - qn * (qn * g_qn).sum(axis, keepdims=True)) / nrm
    return g_q, g_t


def _m2q(R):
    """matrix_to_quaternion (:15-74) for R (..., 9) row-major.  Returns (q, cache).  The candidate with the largest q_abs is
    taken (argmax: the FIRST of equal maxima); for a rotation matrix that q_abs is >= 1, so the 0.1 floor never acts."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = (R[..., i] for i in range(9))
    s = np.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], -1)
    q_abs = np.sqrt(np.maximum(s, 0.0))
    best = np.argmax(q_abs, -1)
    cands = np.stack([np.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
                      np.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
                      np.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
                      np.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)     # (..., 4, 4)
    num = np.take_along_axis(cands, best[..., None, None], -2)[..., 0, :]
    qa = np.take_along_axis(q_abs, best[..., None], -1)
    den = 2.0 * np.maximum(qa, 0.1)
    return num / den, (best, num, qa, den)


# d(candidate numerator)/dR as sparse sign tables: NUM[b][e] = list of (matrix element, sign); element b itself is s_b
_S_SIGNS = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], np.float64)   # d s_b / d (m00, m11, m22)
_NUM = {
    0: {1: ((7, 1), (5, -1)), 2: ((2, 1), (6, -1)), 3: ((3, 1), (1, -1))},
    1: {0: ((7, 1), (5, -1)), 2: ((3, 1), (1, 1)), 3: ((2, 1), (6, 1))},
    2: {0: ((2, 1), (6, -1)), 1: ((3, 1), (1, 1)), 3: ((5, 1), (7, 1))},
    3: {0: ((3, 1), (1, -1)), 1: ((6, 1), (2, 1)), 2: ((7, 1), (5, 1))},
}


def _m2q_backward(cache, g_q):
    """dL/dR (..., 9) from dL/dq: q = num / (2 sqrt(s_b)), num_b = s_b (= q_abs_b^2), the other numerators linear in R."""
    best, num, qa, den = cache
    shp = g_q.shape[:-1]
    gR = np.zeros(shp + (9,), g_q.dtype)
    g_num = g_q / den
    live = (qa[..., 0] > 0.1)                                   # (floor inactive; kept for completeness)
    g_qa = np.where(live, -(g_q * num).sum(-1) / (den[..., 0] ** 2) * 2.0, 0.0)
    flat_b = best.reshape(-1)
    gRf, g_numf, g_qaf, qaf = gR.reshape(-1, 9), g_num.reshape(-1, 4), g_qa.reshape(-1), qa.reshape(-1)
    for b in range(4):
        rows = np.nonzero(flat_b == b)[0]
        if rows.size == 0:
            continue
        # s_b enters through num_b = s_b and through q_abs_b = sqrt(s_b)
        g_s = g_numf[rows, b] + g_qaf[rows] / (2.0 * np.maximum(qaf[rows], 1e-30))
        for d, e in enumerate((0, 4, 8)):
            gRf[rows, e] += _S_SIGNS[b, d] * g_s
        for e, terms in _NUM[b].items():
            for (m, sg) in terms:
                gRf[rows, m] += sg * g_numf[rows, e]
    return gR


def dq2qt(dq, rot_as_q):
    """DQ2QT (:146-165) for dq (N, 8).  Returns (rot (N, 4) | (N, 9), t (N, 3), cache)."""
    real, imag = dq[..., :4], dq[..., 4:]
    rn = np.maximum(np.sqrt((real * real).sum(-1, keepdims=True)), 1e-8)
    r, d = real / rn, imag / rn
    w0, x0, y0, z0 = (r[..., i] for i in range(4))
    w1, x1, y1, z1 = (d[..., i] for i in range(4))
    t = 2 * np.stack([-w1 * x0 + x1 * w0 - y1 * z0 + z1 * y0,
                      -w1 * y0 + x1 * z0 + y1 * w0 - z1 * x0,
                      -w1 * z0 - x1 * y0 + y1 * x0 + z1 * w0], -1)
    R = np.stack([1 - 2 * y0 ** 2 - 2 * z0 ** 2, 2 * x0 * y0 - 2 * w0 * z0, 2 * x0 * z0 + 2 * w0 * y0,
                  2 * x0 * y0 + 2 * w0 * z0, 1 - 2 * x0 ** 2 - 2 * z0 ** 2, 2 * y0 * z0 - 2 * w0 * x0,
                  2 * x0 * z0 - 2 * w0 * y0, 2 * y0 * z0 + 2 * w0 * x0, 1 - 2 * x0 ** 2 - 2 * y0 ** 2], -1)
    if rot_as_q:
        q, mc = _m2q(R)
        return q, t, (r, d, rn, mc)
    return R, t, (r, d, rn, None)


def dq2qt_backward(cache, g_rot, g_t, rot_as_q):
    r, d, rn, mc = cache
    gR = _m2q_backward(mc, g_rot) if rot_as_q else g_rot
    w0, x0, y0, z0 = (r[..., i] for i in range(4))
    w1, x1, y1, z1 = (d[..., i] for i in range(4))
    g = [gR[..., i] for i in range(9)]
    gt = 2 * g_t
    a, b, c = gt[..., 0], gt[..., 1], gt[..., 2]
    # rotation part
    g_w0 = 2 * (-z0 * g[1] + y0 * g[2] + z0 * g[3] - x0 * g[5] - y0 * g[6] + x0 * g[7])
    g_x0 = 2 * (y0 * g[1] + z0 * g[2] + y0 * g[3] - 2 * x0 * g[4] - w0 * g[5] + z0 * g[6] + w0 * g[7] - 2 * x0 * g[8])
    g_y0 = 2 * (-2 * y0 * g[0] + x0 * g[1] + w0 * g[2] + x0 * g[3] + z0 * g[5] - w0 * g[6] + z0 * g[7] - 2 * y0 * g[8])
    g_z0 = 2 * (-2 * z0 * g[0] - w0 * g[1] + x0 * g[2] + w0 * g[3] - 2 * z0 * g[4] + y0 * g[5] + x0 * g[6] + y0 * g[7])
    # translation part: t = 2 (..) bilinear in (r, d)
    g_w0 = g_w0 + x1 * a + y1 * b + z1 * c
    g_x0 = g_x0 - w1 * a - z1 * b + y1 * c
    g_y0 = g_y0 + z1 * a - w1 * b - x1 * c
    g_z0 = g_z0 - y1 * a + x1 * b - w1 * c
    g_w1 = -x0 * a - y0 * b - z0 * c
    g_x1 = w0 * a + z0 * b - y0 * c
    g_y1 = -z0 * a + w0 * b + x0 * c
    g_z1 = y0 * a - x0 * b + w0 * c
    g_r = np.stack([g_w0, g_x0, g_y0, g_z0], -1)
    g_d = np.stack([g_w1, g_x1, g_y1, g_z1], -1)
    # r = real / rn, d = imag / rn, rn = |real| (clamp inactive)
    g_real = (g_r - r * (r * g_r).sum(-1, keepdims=True)) / rn - r * (d * g_d).sum(-1, keepdims=True) / rn
    g_imag = g_d / rn
    return np.concatenate([g_real, g_imag], -1)


def dq_blending(q, t, weights, rot_as_q=True, norm_over_nodes=None):
    """DQBlending (:168-179).  q (K, 4) | (B, K, 4) with B in {1, N}; t likewise; weights (N, K).
    Returns (rot, t_, cache)."""
    if norm_over_nodes is None:
        norm_over_nodes = q.ndim == 3
    q3, t3 = (q[None], t[None]) if q.ndim == 2 else (q, t)
    dq, c1 = qt2dq(q3, t3, norm_over_nodes)
    dq_avg = (dq * weights[..., None]).sum(-2)               # (N, 8)
    rot, t_, c2 = dq2qt(dq_avg, rot_as_q)
    return rot, t_, (c1, c2, dq, weights, rot_as_q, norm_over_nodes, q.ndim == 2)


def dq_blending_backward(cache, g_rot, g_t):
    """(dL/dq, dL/dt, dL/dweights) in the shapes of the inputs."""
    c1, c2, dq, weights, rot_as_q, norm_over_nodes, was2d = cache
    g_avg = dq2qt_backward(c2, g_rot, g_t, rot_as_q)          # (N, 8)
    g_w = (g_avg[..., None, :] * dq).sum(-1)                  # (N, K)
    g_dq = g_avg[..., None, :] * weights[..., None]           # (N, K, 8)
    if dq.shape[0] == 1:
        g_dq = g_dq.sum(0, keepdims=True)
    g_q, g_t_in = qt2dq_backward(c1, g_dq, norm_over_nodes)
    if was2d:
        g_q, g_t_in = g_q[0], g_t_in[0]
    return g_q, g_t_in, g_w


def interpolate(q0, t0, q1, t1, weight, rot_as_q=True):
    """interpolate (:182-187): q0 / q1 (M, 4) 2-D (per-quaternion normalisation), dq0 * weight + dq1 * (1 - weight)."""
    M = q0.shape[0]
    w = np.broadcast_to(np.asarray(weight, q0.dtype).reshape(-1, 1) if np.ndim(weight) else np.full((1, 1), weight, q0.dtype), (M, 1))
    rot, t_, _ = dq_blending(np.stack([q0, q1], 1), np.stack([t0, t1], 1), np.concatenate([w, 1 - w], 1), rot_as_q, norm_over_nodes=False)
    return rot, t_


def quaternion_to_matrix(q):
    """dual_quaternion.py:77-94 (the 2 / |q|^2 form)."""
    r, i, j, k = (q[..., e] for e in range(4))
    two_s = 2.0 / (q * q).sum(-1)
    return np.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1).reshape(q.shape[:-1] + (3, 3))


def transformation_blending(transformations, weights):
    """transformation_blending (:190-197): (K, 4, 4) rigid transforms, weights (N, K) -> (N, 4, 4).  The rotations go through
    matrix_to_quaternion, the blend through DQBlending with a 3-D q (``qs[None]``: node-axis normalisation) and back through
    quaternion_to_matrix."""
    Rs, Ts = transformations[:, :3, :3], transformations[:, :3, 3]
    qs, _ = _m2q(Rs.reshape(-1, 9))
    q, T, _ = dq_blending(qs[None], Ts[None], weights, rot_as_q=True)
    out = np.tile(np.eye(4, dtype=transformations.dtype)[None], (weights.shape[0], 1, 1))
    out[:, :3, :3] = quaternion_to_matrix(q)
    out[:, :3, 3] = T
    return out
