"""ORACLE (test infrastructure, NOT product code).

CPU restatement, in plain torch, of the RigGS skeleton-deformation path and the
pre-rasterizer "render glue".  Every function cites the reference lines it
follows (paths relative to /root/reference).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (riggs_amd/) never does.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
real reference Python in the build container (tests/golden/make_golden.py) and
``tests/test_oracle_deform.py`` checks this restatement against them.

Quaternion convention: (w, x, y, z) everywhere.
"""
from __future__ import annotations

import math
import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# utils/time_utils.py:115-132  quaternion_to_matrix (handles non-unit q)
# --------------------------------------------------------------------------
def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k),
            two_s * (i * j - k * r),
            two_s * (i * k + j * r),
            two_s * (i * j + k * r),
            1 - two_s * (i * i + k * k),
            two_s * (j * k - i * r),
            two_s * (i * k - j * r),
            two_s * (j * k + i * r),
            1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


# --------------------------------------------------------------------------
# utils/time_utils.py:135-205  matrix_to_quaternion (no sign standardisation)
# --------------------------------------------------------------------------
def matrix_to_quaternion(m: torch.Tensor) -> torch.Tensor:
    batch = m.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m.reshape(batch + (9,)), dim=-1)
    arg = torch.stack(
        [1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1
    )
    q_abs = torch.where(arg > 0, torch.sqrt(arg.clamp_min(0)), torch.zeros_like(arg))  # :135-143
    cand = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))  # :196-197
    pick = q_abs.argmax(dim=-1)  # first maximal index, :202-204
    return torch.gather(cand, -2, pick[..., None, None].expand(batch + (1, 4))).squeeze(-2)


# --------------------------------------------------------------------------
# skeleton_utils/skeleton_warp.py:242-273 chain_product_transform, :290-300
# --------------------------------------------------------------------------
def fk_chain(rot_mats: torch.Tensor, joints: torch.Tensor, parents: torch.Tensor):
    """rot_mats (J,3,3), joints (J,3), parents (J,) int; parents[0] ignored.

    Returns posed_joints (J,3), transforms (J,4,4).
    """
    J = joints.shape[0]
    vp = parents.clone().long()
    vp[0] = 0  # :246-247
    c = joints[vp]  # rotation centre = PARENT joint rest position (:249)
    RJ = torch.matmul(rot_mats, c[..., None])[..., 0]
    local_t = c - RJ  # :251
    T = torch.zeros(J, 4, 4, dtype=joints.dtype)
    T[:, :3, :3] = rot_mats
    T[:, :3, 3] = local_t
    T[:, 3, 3] = 1.0
    chain = [T[0]]
    for i in range(1, J):  # :257-263
        chain.append(torch.matmul(chain[int(parents[i])], T[i]))
    G = torch.stack(chain, dim=0)
    jh = F.pad(joints, [0, 1], value=1.0)
    posed = torch.matmul(G, jh[..., None])[..., 0][:, :3]  # :268-271
    return posed, G


# --------------------------------------------------------------------------
# skeleton_warp.py:207-238 bone segment squared distance
# --------------------------------------------------------------------------
def bone_dist2(x: torch.Tensor, joints: torch.Tensor, parents: torch.Tensor) -> torch.Tensor:
    b = joints[1:, :3]
    a = joints[parents[1:].long(), :3]  # :208-209
    p = x[:, None, :3]
    ba = b - a
    denom = torch.clamp((ba * ba).sum(-1, keepdim=True), min=1e-6)  # :226
    t = ((p - a) * ba).sum(-1, keepdim=True) / denom
    t = torch.clamp(t, 0.0, 1.0)  # :228
    s = a + t * ba
    return ((s - p) * (s - p)).sum(-1)  # squared (:233, sqrt=False)


# --------------------------------------------------------------------------
# skeleton_warp.py:41-76 cal_nn_weight_skeleton (gs_kernel=True, no WeightMLP)
# --------------------------------------------------------------------------
def skin_weights(x, joints, parents, node_radius_log, K=-1, weight_offsets=None):
    d2 = bone_dist2(x, joints.detach(), parents)
    if K > 0:
        nn_d2, idx = torch.topk(d2, K, largest=False, dim=1)  # :46-48
        idx = idx + 1
    else:
        nn_d2 = d2
        idx = torch.arange(1, joints.shape[0], dtype=torch.long)[None].expand(d2.shape[0], -1)  # :51-53
    radius = torch.exp(node_radius_log)[idx]  # :65 (child-joint index), time_utils.py:874-876
    w = torch.exp(-nn_d2 / (2 * radius ** 2))  # :66
    if weight_offsets is not None:  # :68-69 (WeightMLP variant)
        w = w * weight_offsets
    w = w + 1e-7  # :71
    w = w / w.sum(dim=-1, keepdim=True)  # :72
    return w, nn_d2, idx


# --------------------------------------------------------------------------
# skeleton_warp.py:130-172 deform_by_pose (LBS-only mode)
# --------------------------------------------------------------------------
def deform_by_pose(x, joints, parents, node_radius_log, local_rot, global_trans, motion_mask, K=-1,
                   template_offsets=None, weight_offsets=None):
    """``weight_offsets`` (N, J-1) = sigmoid(WeightMLP(x)) (:56-69) and ``template_offsets`` (N, 3) = detail_net(x, pose)
    (:152-158) are the outputs of the two optional per-Gaussian heads (evaluated by the caller)."""
    x = x.detach()  # :131
    R = quaternion_to_matrix(local_rot)  # :135
    w, d2, idx = skin_weights(x, joints, parents, node_radius_log, K, weight_offsets)  # :138
    posed, G = fk_chain(R, joints[:, :3], parents)  # :140
    Grot = G[:, :3, :3]
    node_rot = matrix_to_quaternion(Grot.detach())  # :144
    d_nodes = posed + global_trans  # :146
    Gt = G[:, :3, 3]
    Ax = torch.einsum("nkab,nkb->nka", Grot[idx], x[:, None]) + Gt[idx]  # :149
    Ax_avg = (Ax * w[..., None]).sum(dim=1)  # :150
    Ax_avg = Ax_avg + global_trans  # :155-158
    if template_offsets is not None:
        Ax_avg = Ax_avg + template_offsets
    d_xyz = (Ax_avg - x) * motion_mask  # :160-161
    d_rot = (node_rot[idx] * w[..., None]).sum(dim=1) * motion_mask  # :163-164
    d_scaling = torch.zeros(x.shape[0], 3)  # :165
    return {
        "d_xyz": d_xyz, "d_rotation": d_rot, "d_scaling": d_scaling, "d_nodes": d_nodes,
        "nn_idx": idx, "nn_weight": w, "nn_dist2": d2, "local_rotation": local_rot,
        "global_trans": global_trans, "transforms": G, "node_rot": node_rot,
    }


# --------------------------------------------------------------------------
# skeleton_utils/network_utils.py:115-150 PoseMLP + utils/time_utils.py:208-256
# --------------------------------------------------------------------------
def pos_embed(t: torch.Tensor, multires: int) -> torch.Tensor:
    out = [t]
    for f in (2.0 ** torch.linspace(0.0, multires - 1, steps=multires)):
        out.append(torch.sin(t * f))
        out.append(torch.cos(t * f))
    return torch.cat(out, -1)


# --------------------------------------------------------------------------
# gaussian_renderer/__init__.py:74-114 + scene/gaussian_model.py:104-132
# --------------------------------------------------------------------------
def render_glue(xyz, features_dc, features_rest, scaling, rotation, opacity, d_xyz, d_rotation, d_scaling,
                isotropic=False):
    means3D = xyz + d_xyz  # :74
    opac = torch.sigmoid(opacity)  # :79, gm:130-132
    if isotropic:
        sc = torch.exp(scaling[..., :1].repeat(1, 3))  # gm:105-108
    else:
        sc = torch.exp(scaling)
    scales = sc + d_scaling  # :89
    rotations = F.normalize(rotation + d_rotation)  # :90, gm:116-118
    shs = torch.cat((features_dc, features_rest), dim=1)  # gm:124-128
    return means3D, opac, scales, rotations, shs


# --------------------------------------------------------------------------
# utils/graphics_utils.py:42-100 + scene/cameras.py:61-72 camera matrices
# --------------------------------------------------------------------------
def world2view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)  # translate=0, scale=1 (gu:47-52)
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection_matrix(znear, zfar, fovX, fovY) -> torch.Tensor:
    tanY, tanX = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = tanY * znear, tanX * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def projection_matrix_from_K(znear, zfar, K, W, H) -> torch.Tensor:
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    top = znear * cy / fy
    bottom = -znear * (H - cy) / fy
    right = znear * (W - cx) / fx
    left = -znear * cx / fx
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = -(right + left) / (right - left)  # asymmetric sign, gu:94
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_matrices(R, T, fovX, fovY, znear=0.01, zfar=100.0, K=None, W=None, H=None):
    """scene/cameras.py:61-72: returns (world_view_transform, full_proj_transform,
    camera_center), all in the reference's transposed (row-vector) convention."""
    wv = torch.tensor(world2view(R, T)).transpose(0, 1)
    if K is not None:
        proj = projection_matrix_from_K(znear, zfar, K, W, H).transpose(0, 1)
    else:
        proj = projection_matrix(znear, zfar, fovX, fovY).transpose(0, 1)
    full = wv.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
    center = wv.inverse()[3, :3]
    return wv, full, center
