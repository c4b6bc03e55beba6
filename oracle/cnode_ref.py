"""CPU restatement of the stage-1 control-node deformation (second half) and of its gradients.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) — never imported by the product.

* ``cal_nn_weight`` — /root/reference/utils/time_utils.py:934-964 (non-skinning branch): Gaussians and nodes are compared in
  3 + hyper_dim dimensions (xyz of both DETACHED, :944,:947-949; the hyper coordinates of both differentiable); the K nearest
  nodes by squared distance come from pytorch3d.ops.knn_points (third-party, not vendored: PARITY UNPINNED for the search
  itself — restated as "K smallest squared distances, ascending, ties to the lowest index");
  u_k = exp(-d_k / (2 r_k^2)) [* sigmoid(_node_weight_k)], r = exp(_node_radius) (:877-883); w = (u + 1e-7) / sum(u + 1e-7).
* ``forward`` — time_utils.py:1133-1191 for node_trans_bias = None, pred_opacity = pred_color = False, skinning = False:
  local_frame: translate = sum_k w_k (R(local_rot_k + e) (x - n_k) + n_k + trans_k) - x, else sum_k w_k trans_k; times the motion
  mask; rotation = sum_k w_k rot_k * m (d_rot_as_res) or (sum_k w_k (rot_k + e) - e) m + e; scale = sum_k w_k scale_k * m;
  d_nodes = nodes[:, :3] + node_trans.  R(q) is the reference's quaternion_to_matrix (two_s = 2 / |q|^2, time_utils.py:115-132).
* ``backward`` — derived by hand (the reference leaves it to autograd).
Pinned by tests/golden/cnodes_*.npz: outputs and autograd gradients of the reference's own module on CPU
(tests/golden/make_golden.py:fixture_control_nodes).  float64: a checker, not a bit-level model.
"""
import numpy as np

E = np.array([1.0, 0.0, 0.0, 0.0])


def quat_to_mat(q):
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s = 2.0 / (q * q).sum(-1)
    o = np.stack([1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                  s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                  s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)], -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def quat_to_mat_vjp(q, G):
    """dL/dq for R = quat_to_mat(q), G = dL/dR (…, 3, 3): central differences are avoided — analytic via the homogeneous form
    R = I + s * A(q), s = 2/|q|^2, A quadratic in q."""
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    n2 = (q * q).sum(-1)
    s = 2.0 / n2
    A = np.stack([-(j * j + k * k), i * j - k * r, i * k + j * r,
                  i * j + k * r, -(i * i + k * k), j * k - i * r,
                  i * k - j * r, j * k + i * r, -(i * i + j * j)], -1).reshape(q.shape[:-1] + (3, 3))
    g = G
    dA = s[..., None, None] * g  # dL/dA
    dq = np.zeros_like(q)
    dq[..., 0] = (-k * dA[..., 0, 1] + j * dA[..., 0, 2] + k * dA[..., 1, 0] - i * dA[..., 1, 2] - j * dA[..., 2, 0] + i * dA[..., 2, 1])
    dq[..., 1] = (j * dA[..., 0, 1] + k * dA[..., 0, 2] + j * dA[..., 1, 0] - 2 * i * dA[..., 1, 1] - r * dA[..., 1, 2]
                  + k * dA[..., 2, 0] + r * dA[..., 2, 1] - 2 * i * dA[..., 2, 2])
    dq[..., 2] = (-2 * j * dA[..., 0, 0] + i * dA[..., 0, 1] + r * dA[..., 0, 2] + i * dA[..., 1, 0] + k * dA[..., 1, 2]
                  - r * dA[..., 2, 0] + k * dA[..., 2, 1] - 2 * j * dA[..., 2, 2])
    dq[..., 3] = (-2 * k * dA[..., 0, 0] - r * dA[..., 0, 1] + i * dA[..., 0, 2] + r * dA[..., 1, 0] - 2 * k * dA[..., 1, 1]
                  + j * dA[..., 1, 2] + i * dA[..., 2, 0] + j * dA[..., 2, 1])
    ds = (g * A).sum((-1, -2))
    dq += (ds * (-2.0 / n2 ** 2))[..., None] * 2 * q
    return dq


def cal_nn_weight(x, feature, nodes, node_radius_log, node_weight_logit, K, hyper_dim):
    x = np.asarray(x, np.float64)
    nodes = np.asarray(nodes, np.float64)
    if hyper_dim > 0 and feature is not None and np.size(feature):
        xa = np.concatenate([x, np.asarray(feature, np.float64)[:, :hyper_dim]], -1)
        na = nodes
    else:
        xa, na = x, nodes[:, :3]
    d = ((xa[:, None, :] - na[None, :, :]) ** 2).sum(-1)
    idx = np.argsort(d, axis=1, kind="stable")[:, :K]
    dist = np.take_along_axis(d, idx, 1)
    r = np.exp(np.asarray(node_radius_log, np.float64))[idx]
    u = np.exp(-dist / (2 * r * r))
    if node_weight_logit is not None:
        u = u * (1.0 / (1.0 + np.exp(-np.asarray(node_weight_logit, np.float64)[:, 0])))[idx]
    v = u + 1e-7
    return v / v.sum(-1, keepdims=True), dist, idx


def forward(x, feature, mask, nodes, node_radius_log, node_weight_logit, attrs, K, hyper_dim, local_frame, d_rot_as_res):
    x = np.asarray(x, np.float64)
    nodes = np.asarray(nodes, np.float64)
    m = np.asarray(mask, np.float64).reshape(-1, 1)
    w, dist, idx = cal_nn_weight(x, feature, nodes, node_radius_log, node_weight_logit, K, hyper_dim)
    tr, rot, sc = (np.asarray(attrs[k], np.float64) for k in ("d_xyz", "d_rotation", "d_scaling"))
    if local_frame:
        R = quat_to_mat(np.asarray(attrs["local_rotation"], np.float64) + E)
        nn = nodes[idx, :3]
        y = np.einsum("nkab,nkb->nka", R[idx], x[:, None] - nn) + nn + tr[idx]
        translate = ((y * w[..., None]).sum(1) - x) * m
    else:
        translate = (tr[idx] * w[..., None]).sum(1) * m
    if d_rot_as_res:
        rotation = (rot[idx] * w[..., None]).sum(1) * m
    else:
        rotation = (((rot + E)[idx] * w[..., None]).sum(1) - E) * m + E
    scale = (sc[idx] * w[..., None]).sum(1) * m
    return {"d_xyz": translate, "d_rotation": rotation, "d_scaling": scale, "d_nodes": nodes[:, :3] + tr,
            "nn_weight": w, "nn_dist": dist, "nn_idx": idx}


def backward(x, feature, mask, nodes, node_radius_log, node_weight_logit, attrs, K, hyper_dim, local_frame, d_rot_as_res, gout):
    """gout: dict of upstream gradients for d_xyz, d_rotation, d_scaling, d_nodes -> dict of gradients."""
    x = np.asarray(x, np.float64)
    nodes = np.asarray(nodes, np.float64)
    M = nodes.shape[0]
    m = np.asarray(mask, np.float64).reshape(-1, 1)
    w, dist, idx = cal_nn_weight(x, feature, nodes, node_radius_log, node_weight_logit, K, hyper_dim)
    tr, rot, sc = (np.asarray(attrs[k], np.float64) for k in ("d_xyz", "d_rotation", "d_scaling"))
    lq = np.asarray(attrs["local_rotation"], np.float64) + E
    g, h, s_ = (np.asarray(gout[k], np.float64) for k in ("d_xyz", "d_rotation", "d_scaling"))
    gh, hh, sh = g * m, h * m, s_ * m
    rot_eff = rot if d_rot_as_res else rot + E
    if local_frame:
        R = quat_to_mat(lq)
        nn = nodes[idx, :3]
        rel = x[:, None] - nn
        y = np.einsum("nkab,nkb->nka", R[idx], rel) + nn + tr[idx]
    else:
        y = tr[idx]
    dw = (gh[:, None] * y).sum(-1) + (hh[:, None] * rot_eff[idx]).sum(-1) + (sh[:, None] * sc[idx]).sum(-1)
    out = {}
    flat = idx.reshape(-1)
    def scat(vals, width):
        o = np.zeros((M, width))
        np.add.at(o, flat, vals.reshape(-1, width))
        return o
    out["d_xyz"] = scat(w[..., None] * gh[:, None], 3) + np.asarray(gout["d_nodes"], np.float64)
    out["d_rotation"] = scat(w[..., None] * hh[:, None], 4)
    out["d_scaling"] = scat(w[..., None] * sh[:, None], 3)
    if local_frame:
        GR = scat((w[..., None, None] * gh[:, None, :, None] * rel[:, :, None, :]), 9).reshape(M, 3, 3)
        out["local_rotation"] = quat_to_mat_vjp(lq, GR)
    else:
        out["local_rotation"] = np.zeros((M, 4))
    # motion mask
    tsum = (y * w[..., None]).sum(1) - (x if local_frame else 0.0)
    rsum = (rot_eff[idx] * w[..., None]).sum(1) - (0.0 if d_rot_as_res else E)
    ssum = (sc[idx] * w[..., None]).sum(1)
    out["motion_mask"] = ((g * tsum).sum(-1) + (h * rsum).sum(-1) + (s_ * ssum).sum(-1)).reshape(np.asarray(mask).shape)
    # normalisation, kernel
    r = np.exp(np.asarray(node_radius_log, np.float64))[idx]
    e = np.exp(-dist / (2 * r * r))
    if node_weight_logit is not None:
        sg = (1.0 / (1.0 + np.exp(-np.asarray(node_weight_logit, np.float64)[:, 0])))
        nw = sg[idx]
    else:
        nw = np.ones_like(e)
    u = e * nw
    vsum = (u + 1e-7).sum(-1, keepdims=True)
    dv = (dw - (w * dw).sum(-1, keepdims=True)) / vsum
    out["_node_radius"] = scat(dv * u * dist / (r * r), 1)[:, 0]
    if node_weight_logit is not None:
        out["_node_weight"] = scat(dv * e, 1) * (sg * (1 - sg))[:, None]
    dd = dv * u * (-1.0 / (2 * r * r))
    gn = np.zeros_like(nodes)
    gn[:, :3] += np.asarray(gout["d_nodes"], np.float64)
    if hyper_dim > 0 and feature is not None and np.size(feature):
        f = np.asarray(feature, np.float64)
        diff = f[:, None, :hyper_dim] - nodes[idx, 3:]
        gf = np.zeros_like(f)
        gf[:, :hyper_dim] = (dd[..., None] * 2 * diff).sum(1)
        out["feature"] = gf
        gn[:, 3:] = -scat(dd[..., None] * 2 * diff, hyper_dim)
    out["nodes"] = gn
    return out
