"""Independent float64 dense (no tiling, O(N*H*W)) autograd restatement of the
3DGS splatting equations.  It pins oracle/raster_ref.c —
forward AND gradients — for small scenes.  Written with matrix algebra and torch
autograd so that it shares no code and no hand-derived gradient with the C oracle.

Non-differentiable decisions (culling, tile rectangle, alpha/T thresholds, frustum
clamp) are reproduced as constant masks.
"""
import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, d):  # sh (N,16,3), d (N,3) unit
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
             + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
             + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
             + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def quat_R(q):
    r, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def render(means3D, means2D, opac, view, proj, campos, tanx, tany, H, W, bg, shs=None, colors=None, scales=None,
           rots=None, cov6=None, deg=3, mod=1.0):
    """All tensor inputs float64.  view/proj are the transposed (row-vector) 4x4 matrices.
    Returns color (3,H,W), depth (H,W), alpha (H,W), radii (N), n_contrib (H,W)."""
    N = means3D.shape[0]
    dt = means3D.dtype
    ones = torch.ones(N, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    pv = ph @ view  # row-vector convention
    hom = ph @ proj
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None] + means2D[:, :2]
    tz = pv[:, 2]
    if cov6 is None:
        R = quat_R(rots)
        Mm = R * (mod * scales)[:, None, :]
        Sigma = Mm @ Mm.transpose(1, 2)
    else:
        i = [0, 1, 2, 1, 3, 4, 2, 4, 5]
        Sigma = cov6[:, i].reshape(N, 3, 3)
    fx, fy = W / (2 * tanx), H / (2 * tany)
    limx, limy = 1.3 * tanx, 1.3 * tany
    txtz, tytz = pv[:, 0] / tz, pv[:, 1] / tz
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tz).detach(), pv[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tz).detach(), pv[:, 1])
    zero = torch.zeros_like(tz)
    Jm = torch.stack([fx / tz, zero, -fx * tx / tz ** 2, zero, fy / tz, -fy * ty / tz ** 2], -1).reshape(N, 2, 3)
    Wm = view[:3, :3].T  # p_view = Wm p + t
    M2 = Jm @ Wm
    cov2 = M2 @ Sigma @ M2.transpose(1, 2)
    a, b, c = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    A_, B_, C_ = c / det, -b / det, a / det
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    rad = torch.ceil(3 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1) * W - 1) * 0.5
    py = ((ndc[:, 1] + 1) * H - 1) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    x0 = torch.clamp(torch.trunc((px.detach() - rad) / 16), 0, gx)
    x1 = torch.clamp(torch.trunc((px.detach() + rad + 15) / 16), 0, gx)
    y0 = torch.clamp(torch.trunc((py.detach() - rad) / 16), 0, gy)
    y1 = torch.clamp(torch.trunc((py.detach() + rad + 15) / 16), 0, gy)
    vis = (tz.detach() > 0.2) & (det.detach() != 0) & ((x1 - x0) * (y1 - y0) > 0)
    radii = torch.where(vis, rad, torch.zeros_like(rad)).to(torch.int32)
    if colors is None:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh(deg, shs, d) + 0.5, 0.0)
    else:
        rgb = colors
    order = torch.argsort(tz.detach().float(), stable=True)  # fp32 depth bits decide the order
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    tyi, txi = torch.div(ys, 16, rounding_mode="floor"), torch.div(xs, 16, rounding_mode="floor")
    T = torch.ones(H, W, dtype=dt)
    done = torch.zeros(H, W, dtype=torch.bool)
    col = torch.zeros(3, H, W, dtype=dt)
    dep = torch.zeros(H, W, dtype=dt)
    alp = torch.zeros(H, W, dtype=dt)
    count = torch.zeros(H, W, dtype=torch.int64)
    ncontrib = torch.zeros(H, W, dtype=torch.int64)
    for k in order.tolist():
        if not bool(vis[k]):
            continue
        intile = (txi >= x0[k]) & (txi < x1[k]) & (tyi >= y0[k]) & (tyi < y1[k])
        live = intile & ~done
        count = count + live.to(torch.int64)
        dx, dy = px[k] - xs, py[k] - ys
        power = -0.5 * (A_[k] * dx * dx + C_[k] * dy * dy) - B_[k] * dx * dy
        G = torch.exp(power)
        # min(0.99, o*G) with the gradient still flowing when capped (Appendix B / upstream behaviour)
        raw = opac[k] * G
        alpha = torch.where(raw > 0.99, raw - (raw - 0.99).detach(), raw)
        ok = live & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
        testT = T * (1 - alpha)
        stop = ok & (testT.detach() < 1e-4)
        done = done | stop
        use = ok & ~stop
        w = torch.where(use, alpha * T, torch.zeros_like(T))
        col = col + rgb[k][:, None, None] * w
        dep = dep + tz[k] * w
        alp = alp + w
        T = torch.where(use, testT, T)
        ncontrib = torch.where(use, count, ncontrib)
    col = col + T * bg[:, None, None]
    return col, dep, alp, radii, ncontrib, T
