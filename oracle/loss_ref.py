"""CPU restatement of the trainer's image loss and of its gradient.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) — never imported by the product.

* ``l1`` — /root/reference/utils/loss_utils.py:17-18.
* ``ssim`` — loss_utils.py:33-77: 11-tap Gaussian (sigma 1.5, float32, normalised) as a 2-D outer-product window, zero padding
  (``padding=window_size // 2``), per-channel (``groups=channel``), C1 = 0.01^2, C2 = 0.03^2, mean over every element.
* ``grad`` — d(a * l1 + b * ssim)/d(image), derived by hand (the reference leaves it to autograd):
  with the five windowed moments mu1, mu2, E11, E22, E12 and ssim = A B / (C D),
    d/dmu1 = 2 mu2 (B - A)/(C D) - ssim 2 mu1 (D - C)/(C D),   d/dE11 = -ssim / D,   d/dE12 = 2 A / (C D),
    d/dx   = [ G*dmu1 + 2 x (G*dE11) + y (G*dE12) ] / n   (G symmetric, zero padded)   and   d l1/dx = sign(x - y)/n.
Pinned by tests/golden/loss_l1_ssim.npz: values and autograd gradients of the reference's own functions on CPU
(tests/golden/make_golden.py:fixture_loss).  float64 arithmetic: a checker, not a bit-level model of conv2d.
"""
import math

import numpy as np


def window1d():
    g = np.array([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=np.float32)
    return (g / g.sum(dtype=np.float32)).astype(np.float32)


def _conv(img, w2):
    """Zero-padded 11x11 correlation per channel; img (C, H, W) float64."""
    C, H, W = img.shape
    p = np.zeros((C, H + 10, W + 10))
    p[:, 5:5 + H, 5:5 + W] = img
    out = np.zeros_like(img)
    for dy in range(11):
        for dx in range(11):
            out += w2[dy, dx] * p[:, dy:dy + H, dx:dx + W]
    return out


def _moments(x, y):
    g = window1d()
    w2 = np.outer(g, g).astype(np.float32).astype(np.float64)  # _1D_window.mm(_1D_window.t()).float()
    return w2, _conv(x, w2), _conv(y, w2), _conv(x * x, w2), _conv(y * y, w2), _conv(x * y, w2)


def l1(x, y):
    return float(np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64)).mean())


def ssim(x, y):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    _, m1, m2, e11, e22, e12 = _moments(x, y)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    s1, s2, s12 = e11 - m1 * m1, e22 - m2 * m2, e12 - m1 * m2
    return float((((2 * m1 * m2 + C1) * (2 * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s1 + s2 + C2))).mean())


def grad(x, y, g_l1, g_ssim):
    """d(g_l1 * l1 + g_ssim * ssim)/dx."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    w2, m1, m2, e11, e22, e12 = _moments(x, y)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    s1, s2, s12 = e11 - m1 * m1, e22 - m2 * m2, e12 - m1 * m2
    A, B, C, D = 2 * m1 * m2 + C1, 2 * s12 + C2, m1 * m1 + m2 * m2 + C1, s1 + s2 + C2
    inv = 1.0 / (C * D)
    s = A * B * inv
    dmu = 2 * m2 * (B - A) * inv - s * 2 * m1 * (D - C) * inv
    de11 = -s / D
    de12 = 2 * A * inv
    n = x.size
    return g_l1 * np.sign(x - y) / n + g_ssim / n * (_conv(dmu, w2) + 2 * x * _conv(de11, w2) + y * _conv(de12, w2))


# ------------------------------------------------------------------------------------------------------------------------
# Skeleton projection loss (second half): TrainRig.cal_skeleton_loss, train_rig.py:309-314.
#
# * ``sampling_steps`` / ``sampling_skeleton_points`` — train_rig.py:264-276: with d_k the (detached) bone lengths,
#   S = int(max d / (sum d / 512)) equally spaced parameters t in [0, 1] (torch.linspace, float32) on EVERY bone;
#   point (s, k) = t_s * child_k + (1 - t_s) * parent_k, flattened s-major.
# * ``project_nodes`` — utils/other_utils.py:101-127: row-vector world->view transform, pinhole with
#   fx = W / (2 tan(FoVx/2)), fy likewise, principal point K[0,2], K[1,2] or the image centre; elements are (y, x).
# * ``chamfer_l1`` — pytorch3d.loss.chamfer_distance(x, y, norm=1) with its defaults.  pytorch3d is a pip dependency of
#   the reference (not vendored, not installed here): PARITY UNPINNED for this factor — restated from its published
#   definition: per point the L1 distance to the nearest point of the other set (ties: lowest index), mean per set, the
#   two directions added.  d|a - b|/da = sign(a - b) with sign(0) = 0.
# Pinned (everything but the chamfer factor) by tests/golden/skelproj_*.npz: the reference's own sampling, projection and
# composition with autograd's gradient (tests/golden/make_golden.py:fixture_skeleton_projection).
# ------------------------------------------------------------------------------------------------------------------------
def sampling_steps(nodes, parents, num_sample=512):
    """The linspace parameters (float32, as torch computes them: from both ends towards the middle)."""
    nodes = np.asarray(nodes, np.float32)
    par = np.asarray(parents)[1:].astype(np.int64)
    d = np.linalg.norm((nodes[1:] - nodes[par]).astype(np.float32), axis=-1).astype(np.float32)
    each = np.float32(d.sum(dtype=np.float32) / np.float32(num_sample))
    S = int(np.float32(d.max()) / each)
    if S <= 0:
        return np.zeros(0, np.float32)
    if S == 1:
        return np.zeros(1, np.float32)
    step = np.float32(1.0) / np.float32(S - 1)
    i = np.arange(S)
    lo = (np.float32(0.0) + step * i.astype(np.float32)).astype(np.float32)
    hi = (np.float32(1.0) - step * (S - 1 - i).astype(np.float32)).astype(np.float32)
    return np.where(i < S // 2, lo, hi).astype(np.float32)


def sampling_skeleton_points(nodes, parents, t):
    nodes = np.asarray(nodes, np.float64)
    par = np.asarray(parents)[1:].astype(np.int64)
    t = np.asarray(t, np.float64)[:, None, None]
    return (t * nodes[1:][None] + (1.0 - t) * nodes[par][None]).reshape(-1, 3)


def intrinsics(FoVx, FoVy, H, W, K=None):
    fx = W / (2.0 * math.tan(FoVx * 0.5))
    fy = H / (2.0 * math.tan(FoVy * 0.5))
    if K is not None and np.size(K):
        return fx, fy, float(K[0][2]), float(K[1][2])
    return fx, fy, W / 2.0, H / 2.0


def project_nodes(points, world_view_transform, fx, fy, cx, cy):
    V = np.asarray(world_view_transform, np.float64)
    tr = np.asarray(points, np.float64) @ V[:3, :3] + V[3, :3]
    return np.stack([fy * tr[:, 1] / tr[:, 2] + cy, fx * tr[:, 0] / tr[:, 2] + cx], -1)


def chamfer_l1(x, y):
    """-> (loss, nearest index in y per x, nearest index in x per y)."""
    d = np.abs(np.asarray(x, np.float64)[:, None, :] - np.asarray(y, np.float64)[None, :, :]).sum(-1)
    ix, iy = d.argmin(1), d.argmin(0)
    return d.min(1).mean() + d.min(0).mean(), ix, iy


def skeleton_projection_loss(nodes, parents, world_view_transform, fx, fy, cx, cy, thinned, t=None):
    """-> (loss, d loss / d nodes)."""
    nodes = np.asarray(nodes, np.float64)
    par = np.asarray(parents)[1:].astype(np.int64)
    if t is None:
        t = sampling_steps(nodes, parents)
    t = np.asarray(t, np.float64)
    pts = sampling_skeleton_points(nodes, parents, t)
    V = np.asarray(world_view_transform, np.float64)
    tr = pts @ V[:3, :3] + V[3, :3]
    proj = np.stack([fy * tr[:, 1] / tr[:, 2] + cy, fx * tr[:, 0] / tr[:, 2] + cx], -1)
    thinned = np.asarray(thinned, np.float64)
    loss, ix, iy = chamfer_l1(proj, thinned)
    P, M = proj.shape[0], thinned.shape[0]
    gp = np.sign(proj - thinned[ix]) / P
    np.add.at(gp, iy, np.sign(proj[iy] - thinned) / M)
    # back through the pinhole: element 0 is y, element 1 is x
    gtr = np.zeros_like(tr)
    gtr[:, 1] = gp[:, 0] * fy / tr[:, 2]
    gtr[:, 0] = gp[:, 1] * fx / tr[:, 2]
    gtr[:, 2] = -(gp[:, 0] * fy * tr[:, 1] + gp[:, 1] * fx * tr[:, 0]) / tr[:, 2] ** 2
    gpts = (gtr @ V[:3, :3].T).reshape(t.shape[0], -1, 3)
    g = np.zeros_like(nodes)
    np.add.at(g, np.arange(1, nodes.shape[0]), (t[:, None, None] * gpts).sum(0))
    np.add.at(g, par, ((1.0 - t)[:, None, None] * gpts).sum(0))
    return loss, g


# ---- the stage-2 objective's two regularisers (train_rig.py:446-456, 474-482) -----------------------------------------------------
def stage2_regularisers(template_offsets, local_rotation, is_template, lambda_template_offsets=1.0, lambda_template_fixed=100.0):
    """The terms ``render_and_cal_loss`` adds to the image loss once the heads are on, and their gradients (float64).

    * train_rig.py:446-456 — ``l2_loss(template_offsets, 0)`` = mean over ALL N x 3 entries (utils/loss_utils.py:29-30), weighted
      ``lambda_template_offsets``, x1e3 when ``viewpoint_cam.uid == template_idx``;
    * train_rig.py:474-482 — on the template camera only: ``lambda_template_fixed * l2_loss(local_rotation, (1,0,0,0))``, the
      mean over the J x 4 entries.
    Pinned by tests/golden/objective_tree8_n48.npz: the reference's own method run on a bare TrainRig
    (tests/golden/make_objective_golden.py), both cameras."""
    T = np.asarray(template_offsets, np.float64)
    q = np.asarray(local_rotation, np.float64).reshape(-1, 4)
    lam_t = float(lambda_template_offsets) * (1e3 if is_template else 1.0)
    t_loss = float((T ** 2).mean())
    out = {"template_offsets_loss": t_loss, "lambda_template_offsets": lam_t, "g_template_offsets": lam_t * 2.0 * T / T.size,
           "total": lam_t * t_loss, "template_fixed_loss": None, "g_local_rotation": np.zeros_like(q)}
    if is_template and lambda_template_fixed > 1e-8:
        d = q - np.array([1.0, 0.0, 0.0, 0.0])
        f_loss = float((d ** 2).mean())
        out["template_fixed_loss"] = f_loss
        out["g_local_rotation"] = float(lambda_template_fixed) * 2.0 * d / d.size
        out["total"] += float(lambda_template_fixed) * f_loss
    return out
