"""Stage-1 control-node deformation with the reference's names (/root/reference/utils/time_utils.py:770-1236
``ControlNodeWarp``): K nearest control nodes per Gaussian in (xyz, hyper) space, Gaussian-kernel weights, blend of the nodes'
translations / local-frame rotations / rotation and scale residuals — ONE HIP launch forward, two backward (csrc/cnode.hip)
instead of a KNN extension call, ~10 (N, K, ·) gathers, an einsum and autograd's replay of them.

Covered: the configuration the trainer ships (KNN weights; ``local_frame``, ``d_rot_as_res``, ``with_node_weight``, ``hyper_dim``
free) through the HIP kernels; ``pred_opacity`` / ``pred_color`` (two more blends with the same weights, torch ops on the
kernel's neighbour lists) and ``skinning=True`` (softmax of a per-Gaussian (N, M) feature instead of KNN weights: a dense
(N, M) x (M, 14) product, left to the GEMM library) as the reference defines them; ``node_trans_bias`` (the GUI's drag-to-edit
path, time_utils.py:1165-1213: an as-rigid-as-possible re-posing of the Gaussians around dragged nodes, under ``no_grad``) in
torch ops on top of the kernel's blend — it runs per mouse event, not per training step.  The node network
(``self.network``: nodes, t -> per-node attributes; 512-1024 rows, time_utils.py:990-1002) stays a torch module supplied by the
caller — it is a few hundred rows through an MLP, not a per-Gaussian cost.  No CPU / eager fallback.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib as L

LOCAL_FRAME, ROT_AS_RES = 1, 2


class _ControlNodeBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, feature, mask, nodes, radius_log, weight_logit, trans, rot, scale, local_rot, K, hyper, flags):
        N, M = x.shape[0], nodes.shape[0]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        d_xyz, d_rot, d_scale = torch.empty(N, 3, **f32), torch.empty(N, 4, **f32), torch.empty(N, 3, **f32)
        nn_idx = torch.empty(N, K, dtype=torch.int32, device=dev)
        nn_weight, nn_dist = torch.empty(N, K, **f32), torch.empty(N, K, **f32)
        fs = feature.shape[1] if feature is not None else 0
        L.check(L.lib().riggs_cnode_forward(N, M, K, hyper, fs, nodes.shape[1], flags, x.data_ptr(), L.ptr(feature), L.ptr(mask),
                                            nodes.data_ptr(), radius_log.data_ptr(), L.ptr(weight_logit), trans.data_ptr(),
                                            rot.data_ptr(), scale.data_ptr(), L.ptr(local_rot), d_xyz.data_ptr(),
                                            d_rot.data_ptr(), d_scale.data_ptr(), nn_idx.data_ptr(), nn_weight.data_ptr(),
                                            nn_dist.data_ptr(), L.stream_ptr()), "riggs_cnode_forward")
        ctx.save_for_backward(x, feature, mask, nodes, radius_log, weight_logit, trans, rot, scale, local_rot, nn_idx, nn_dist)
        ctx.cfg = (K, hyper, flags)
        ctx.mark_non_differentiable(nn_idx, nn_weight, nn_dist)
        ctx.set_materialize_grads(False)
        return d_xyz, d_rot, d_scale, nn_idx, nn_weight, nn_dist

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_scale, *_):
        x, feature, mask, nodes, radius_log, weight_logit, trans, rot, scale, local_rot, nn_idx, nn_dist = ctx.saved_tensors
        K, hyper, flags = ctx.cfg
        N, M = x.shape[0], nodes.shape[0]
        lib = L.lib()
        f32 = dict(dtype=torch.float32, device=x.device)
        c = lambda g: None if g is None else g.to(torch.float32).contiguous()  # noqa: E731
        g_xyz, g_rot, g_scale = c(g_xyz), c(g_rot), c(g_scale)
        need = ctx.needs_input_grad
        g_feature = torch.empty_like(feature) if (feature is not None and need[1]) else None
        g_mask = torch.empty_like(mask) if (mask is not None and need[2]) else None
        g_trans, g_nrot, g_nscale = torch.empty_like(trans), torch.empty_like(rot), torch.empty_like(scale)
        g_local = torch.empty_like(local_rot) if local_rot is not None else None
        g_radius = torch.empty_like(radius_log)
        g_weight = torch.empty_like(weight_logit) if weight_logit is not None else None
        g_hyper = torch.empty(M, hyper, **f32) if hyper > 0 else None
        ws = torch.empty(max(int(lib.riggs_cnode_backward_workspace_floats(N, M, K, hyper)), 1), **f32)
        fs = feature.shape[1] if feature is not None else 0
        L.check(lib.riggs_cnode_backward(N, M, K, hyper, fs, nodes.shape[1], flags, x.data_ptr(), L.ptr(feature), L.ptr(mask),
                                         nodes.data_ptr(), radius_log.data_ptr(), L.ptr(weight_logit), trans.data_ptr(),
                                         rot.data_ptr(), scale.data_ptr(), L.ptr(local_rot), nn_idx.data_ptr(), nn_dist.data_ptr(),
                                         L.ptr(g_xyz), L.ptr(g_rot), L.ptr(g_scale), L.ptr(g_feature), L.ptr(g_mask),
                                         g_trans.data_ptr(), g_nrot.data_ptr(), g_nscale.data_ptr(), L.ptr(g_local),
                                         g_radius.data_ptr(), L.ptr(g_weight), L.ptr(g_hyper), ws.data_ptr(), L.stream_ptr()),
                "riggs_cnode_backward")
        g_nodes = None
        if need[3]:
            g_nodes = torch.zeros_like(nodes)
            if hyper > 0:
                g_nodes[:, 3:3 + hyper] = g_hyper
        return (None, g_feature, g_mask, g_nodes, g_radius, g_weight, g_trans, g_nrot, g_nscale, g_local, None, None, None)


def control_node_blend(x, feature, motion_mask, nodes, _node_radius, _node_weight, node_attrs, K=3, hyper_dim=0,
                       local_frame=False, d_rot_as_res=True):
    """The per-Gaussian part of ``ControlNodeWarp.forward`` (time_utils.py:1138-1191) given the nodes' attributes
    ``node_attrs = {'d_xyz', 'd_rotation', 'd_scaling', 'local_rotation'}``; returns the reference's dict (without ``d_nodes``)
    plus ``nn_idx`` (int32), ``nn_weight``, ``nn_dist`` of ``cal_nn_weight`` (:934-964)."""
    x = L.require_cuda_f32("x", x.detach(), (x.shape[0], 3)).contiguous()
    nodes = L.require_cuda_f32("nodes", nodes).contiguous()
    M = nodes.shape[0]
    hyper = hyper_dim if (hyper_dim > 0 and feature is not None) else 0
    if feature is not None:
        feature = L.require_cuda_f32("feature", feature).contiguous()
        if hyper == 0:
            feature = None
    mask = None
    N = x.shape[0]
    if isinstance(motion_mask, torch.Tensor):
        if motion_mask.numel() == N:
            mask = L.require_cuda_f32("motion_mask", motion_mask).reshape(N).contiguous()
        elif motion_mask.numel() != 1 or float(motion_mask) != 1.0:
            raise NotImplementedError("motion_mask must be per Gaussian (N, 1) or 1")
    elif motion_mask is not None and float(motion_mask) != 1.0:
        raise NotImplementedError("motion_mask must be per Gaussian (N, 1) or 1")
    f = lambda name, t, w: L.require_cuda_f32(name, t, (M, w)).contiguous()  # noqa: E731
    trans, rot, scale = f("d_xyz", node_attrs["d_xyz"], 3), f("d_rotation", node_attrs["d_rotation"], 4), f("d_scaling", node_attrs["d_scaling"], 3)
    local_rot = f("local_rotation", node_attrs["local_rotation"], 4) if local_frame else None
    radius = L.require_cuda_f32("_node_radius", _node_radius, (M,)).contiguous()
    weight = L.require_cuda_f32("_node_weight", _node_weight).reshape(M).contiguous() if _node_weight is not None else None
    flags = (LOCAL_FRAME if local_frame else 0) | (ROT_AS_RES if d_rot_as_res else 0)
    d_xyz, d_rot, d_scale, nn_idx, nn_weight, nn_dist = _ControlNodeBlend.apply(
        x, feature, mask, nodes, radius, weight, trans, rot, scale, local_rot, int(K), int(hyper), flags)
    return {"d_xyz": d_xyz, "d_rotation": d_rot, "d_scaling": d_scale, "d_opacity": None, "d_color": None,
            "nn_idx": nn_idx, "nn_weight": nn_weight, "nn_dist": nn_dist}


def knn_weights_torch(x, feature, nodes, _node_radius, _node_weight, nn_idx, hyper_dim):
    """``cal_nn_weight`` (time_utils.py:934-964) as differentiable torch ops on GIVEN neighbour lists (the HIP kernel's): used
    where autograd has to see the weights themselves (the opacity / colour blends)."""
    idx = nn_idx.long()
    q, nd = x.detach(), nodes[..., :3].detach()
    if hyper_dim > 0 and feature is not None:
        q = torch.cat([q, feature[..., :hyper_dim]], dim=-1)
        nd = torch.cat([nd, nodes[..., 3:3 + hyper_dim]], dim=-1)
    d2 = ((q[:, None] - nd[idx]) ** 2).sum(-1)
    w = torch.exp(-d2 / (2 * torch.exp(_node_radius)[idx] ** 2))
    if _node_weight is not None:
        w = w * torch.sigmoid(_node_weight)[idx][..., 0]
    w = w + 1e-7
    return w / w.sum(dim=-1, keepdim=True)


def skinning_blend(feature, motion_mask, node_attrs, d_rot_as_res=True, pred_opacity=False, pred_color=False):
    """``ControlNodeWarp.forward`` with ``skinning=True`` (time_utils.py:934-938, 1160-1191, 1214-1225; ``local_frame`` off — the
    reference's einsum does not take the skinning weights): weights = softmax over ALL nodes of the per-Gaussian feature, every
    blend a dense (N, M) x (M, c) product."""
    w = torch.softmax(feature, dim=-1)
    rot_bias = torch.tensor([1.0, 0.0, 0.0, 0.0], device=feature.device)
    mm = motion_mask
    out = {"d_xyz": (w @ node_attrs["d_xyz"]) * mm, "d_scaling": (w @ node_attrs["d_scaling"]) * mm}
    if d_rot_as_res:
        out["d_rotation"] = (w @ node_attrs["d_rotation"]) * mm
    else:
        out["d_rotation"] = ((w @ (node_attrs["d_rotation"] + rot_bias)) - rot_bias) * mm + rot_bias
    out["d_opacity"] = (w @ node_attrs["d_opacity"]) * mm if pred_opacity else None
    out["d_color"] = (w @ node_attrs["d_color"]) * mm if pred_color else None
    out["nn_weight"], out["nn_idx"], out["nn_dist"] = w, torch.arange(w.shape[1], device=w.device), None
    return out


# ---- the editing path's geometry (all under no_grad; M nodes, a few hundred) -------------------------------------------------
def _quat_to_mat(q):
    """Rotation matrices of un-normalised (w, x, y, z) quaternions, scale 2 / |q|^2 (time_utils.py:115-132)."""
    w, a, b, c = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    rows = [1 - s * (b * b + c * c), s * (a * b - c * w), s * (a * c + b * w),
            s * (a * b + c * w), 1 - s * (a * a + c * c), s * (b * c - a * w),
            s * (a * c - b * w), s * (b * c + a * w), 1 - s * (a * a + b * b)]
    return torch.stack(rows, -1).reshape(q.shape[:-1] + (3, 3))


def _mat_to_quat(m):
    """(w, x, y, z) of rotation matrices: of the four algebraically equal candidates the one with the largest denominator
    (time_utils.py:146-205; no sign convention)."""
    sq = torch.stack([1.0 + m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2], 1.0 + m[..., 0, 0] - m[..., 1, 1] - m[..., 2, 2],
                      1.0 - m[..., 0, 0] + m[..., 1, 1] - m[..., 2, 2], 1.0 - m[..., 0, 0] - m[..., 1, 1] + m[..., 2, 2]], -1)
    qa = torch.where(sq > 0, sq.clamp_min(0).sqrt(), torch.zeros_like(sq))
    a, b, c = m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1]
    e, f, g = m[..., 1, 0] + m[..., 0, 1], m[..., 0, 2] + m[..., 2, 0], m[..., 1, 2] + m[..., 2, 1]
    cand = torch.stack([torch.stack([qa[..., 0] ** 2, a, b, c], -1), torch.stack([a, qa[..., 1] ** 2, e, f], -1),
                        torch.stack([b, e, qa[..., 2] ** 2, g], -1), torch.stack([c, f, g, qa[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * qa[..., None].clamp_min(0.1))
    pick = qa.argmax(-1)
    return torch.gather(cand, -2, pick[..., None, None].expand(pick.shape + (1, 4))).squeeze(-2)


def _quat_mul(p, q):
    """Hamilton product, result with a non-negative real part (time_utils.py:99-113)."""
    pw, px, py, pz = p.unbind(-1)
    qw, qx, qy, qz = q.unbind(-1)
    r = torch.stack([pw * qw - px * qx - py * qy - pz * qz, pw * qx + px * qw + py * qz - pz * qy,
                     pw * qy - px * qz + py * qw + pz * qx, pw * qz + px * qy - py * qx + pz * qw], -1)
    return torch.where(r[..., :1] < 0, -r, r)


def _knn_sq(a, b, K, rows=32768):
    """The K nearest rows of ``b`` for every row of ``a`` by squared Euclidean distance, ascending (what the reference asks
    pytorch3d.ops.knn_points for); exact differences, in slabs of rows."""
    ds, ix = [], []
    for s0 in range(0, a.shape[0], rows):
        d = ((a[s0:s0 + rows, None, :] - b[None, :, :]) ** 2).sum(-1)
        dk, ik = d.topk(K, dim=1, largest=False, sorted=True)
        ds.append(dk)
        ix.append(ik)
    return torch.cat(ds), torch.cat(ix)


class StaticNodeNetwork(nn.Module):
    """``StaticNetwork(return_tensors=True)`` (time_utils.py:288-301): zero attributes for every node."""

    def forward(self, x, t, **kwargs):
        z3, z4 = torch.zeros_like(x), torch.zeros(x.shape[0], 4, dtype=x.dtype, device=x.device)
        return {"d_xyz": z3, "d_rotation": z4, "d_scaling": z3.clone(), "local_rotation": z4.clone(), "hidden": None,
                "d_opacity": None, "d_color": None}


class ControlNodeWarp(nn.Module):
    """``ControlNodeWarp`` (time_utils.py:770-1236) for the shipped configuration: parameters ``nodes`` (M, 3 + hyper_dim),
    ``_node_radius`` (M), ``_node_weight`` (M, 1) with the reference's names and activations; ``network`` is the node network
    (a torch module ``(nodes_xyz, t) -> dict``; default: the static one)."""

    def __init__(self, node_num=512, K=3, with_node_weight=True, local_frame=False, d_rot_as_res=True, hyper_dim=2, network=None,
                 pred_opacity=False, pred_color=False, skinning=False, **kwargs):
        super().__init__()
        if skinning and local_frame:
            raise NotImplementedError("skinning with local_frame: the reference's own forward fails there (its einsum expects per-"
                                      "Gaussian neighbour lists, time_utils.py:1152)")
        self.skinning, self.pred_opacity, self.pred_color = bool(skinning), bool(pred_opacity), bool(pred_color)
        hyper_dim = 0 if skinning else hyper_dim  # "skinning should not be with hyper" (time_utils.py:782)
        self.K, self.with_node_weight, self.local_frame, self.d_rot_as_res, self.hyper_dim = K, with_node_weight, local_frame, d_rot_as_res, hyper_dim
        self.network = network if network is not None else StaticNodeNetwork()
        self.nodes = nn.Parameter(torch.randn(node_num, 3 + hyper_dim))
        if not skinning:  # (time_utils.py:807-810)
            self._node_radius = nn.Parameter(torch.randn(node_num))
            if with_node_weight:
                self._node_weight = nn.Parameter(torch.zeros(node_num, 1))
        self.reg_loss = 0.

    @property
    def node_radius(self):
        return torch.exp(self._node_radius)

    @property
    def node_weight(self):
        return torch.sigmoid(self._node_weight)

    @property
    def node_num(self):
        return self.nodes.shape[0]

    def trainable_parameters(self):
        if self.skinning:  # (time_utils.py:825-827)
            return [{"params": list(self.network.parameters()), "name": "deform"}, {"params": [self.nodes], "name": "nodes"}]
        node_params = [self.nodes, self._node_radius] + ([self._node_weight] if self.with_node_weight else [])
        return [{"params": list(self.network.parameters()), "name": "deform"}, {"params": node_params, "name": "nodes"}]

    def expand_time(self, t):
        return t.unsqueeze(0).expand(self.nodes.shape[0], -1)

    def node_deform(self, t, **kwargs):
        if t.dim() == 3:  # (M, T, 1): the nodes at T times each (time_utils.py:990-1002)
            M, T = t.shape[0], t.shape[1]
            x = self.nodes[:, None, :3].detach().expand(M, T, 3).reshape(-1, 3)
            v = self.network(x=x, t=t.reshape(-1, 1), **kwargs)
            return {k: (a.view(M, T, a.shape[-1]) if a is not None else None) for k, a in v.items()}
        return self.network(x=self.nodes[..., :3].detach(), t=t, **kwargs)

    # ---- the editing path (node_trans_bias): time_utils.py:1004-1011, 1044-1077, 969-988, 1122-1131, 1165-1213 -------------
    def get_trajectory(self, t_samp_num=8):
        t = torch.linspace(0, 1, t_samp_num, device=self.nodes.device)[None, :, None].expand(self.node_num, t_samp_num, 1)
        nd = self.node_deform(t=t)
        traj = self.nodes[:, None, :3].detach() + nd["d_xyz"]
        return traj.detach(), {k: (v[:, 0] if v is not None else None) for k, v in nd.items()}

    def geodesic_distance_floyd(self, cur_node, K=8):
        """All-pairs shortest paths over the K-nearest-neighbour graph of the nodes (Floyd-Warshall on an (M, M) matrix)."""
        M = cur_node.shape[0]
        d2, idx = _knn_sq(cur_node, cur_node, K + 1)
        dist = torch.full((M, M), float("inf"), device=cur_node.device)
        dist.scatter_(1, idx, d2.sqrt())
        dist = torch.minimum(dist, dist.T)
        for i in range(M):
            dist = torch.minimum(dist[:, i, None] + dist[None, i, :], dist)
        return dist

    def cal_nn_weight_floyd(self, x, t0, cur_node, K=None, GraphK=2, temperature=1.0, cache_name="floyd", XisNode=False):
        """Weights of the K graph-nearest nodes of every point's nearest node; the graph distances are cached per name until
        the time moves by more than 1e-2 (time_utils.py:969-988)."""
        if not hasattr(self, cache_name + "_nn_dist") or (t0 is not None and (getattr(self, cache_name + "_t") - t0).abs().max() > 1e-2):
            gd, gi = self.geodesic_distance_floyd(cur_node=cur_node, K=GraphK).sort(dim=1)
            o = 1 if XisNode else 0
            setattr(self, cache_name + "_nn_dist", gd[:, o:K + o])
            setattr(self, cache_name + "_nn_idx", gi[:, o:K + o])
            if t0 is not None:
                setattr(self, cache_name + "_t", t0.clone())
        d2, i1 = _knn_sq(x, cur_node, 1)
        d2, i1 = d2[:, 0], i1[:, 0]
        kd = getattr(self, cache_name + "_nn_dist")[i1] + d2[:, None]
        return torch.softmax(-kd / temperature, dim=-1), kd, getattr(self, cache_name + "_nn_idx")[i1]

    def p2dR(self, p, p0=None, K=8, as_quat=True, mode="trajectory", t0=None):
        """Per node the rotation that best maps its edges to its K neighbours at rest (``p0``) onto the edges after the drag
        (``p``): Kabsch on weighted unit edges (time_utils.py:1044-1077).  Neighbours: K nearest by the nodes' trajectories
        (four time samples), by graph distance (``floyd``) or by rest position."""
        p = p.detach()
        nodes = self.nodes[..., :3].detach()
        t0_deform = None
        if mode == "trajectory":
            traj, t0_deform = self.get_trajectory(t_samp_num=4)
            base = traj[:, 0] if p0 is None else p0
            flat = traj.reshape(traj.shape[0], -1)
            d2, idx = _knn_sq(flat, flat, K + 1)
            d2, idx = d2[:, 1:], idx[:, 1:]
            w = torch.softmax(d2 / d2.mean(), dim=-1)
            edges = base[idx] - base[:, None]
        elif mode == "floyd":
            w, _, idx = self.cal_nn_weight_floyd(x=p, t0=t0, cur_node=p0, K=K + 1, GraphK=4, temperature=1e-1, cache_name="p2dR", XisNode=True)
            w, idx = w[:, 1:], idx[:, 1:]
            edges = p0[idx] - p0[:, None]
        else:
            d2, idx = _knn_sq(nodes, nodes, K + 1)
            d2, idx = d2[:, 1:], idx[:, 1:]
            w = torch.softmax(d2 / d2.mean(), dim=-1)
            base = nodes if p0 is None else p0
            edges = base[idx] - base[:, None]
        edges_t = p[idx] - p[:, None]
        edges = edges / (edges.norm(dim=-1, keepdim=True) + 1e-5)
        edges_t = edges_t / (edges_t.norm(dim=-1, keepdim=True) + 1e-5)
        S = torch.einsum("nka,nk,nkb->nab", edges, w, edges_t)
        U, _, Vh = torch.linalg.svd(S.cpu())  # (M, 3, 3): tiny, and the host's LAPACK is what torch.svd means everywhere
        dR = (Vh.transpose(-1, -2) @ U.transpose(-1, -2)).to(S.device)
        return (_mat_to_quat(dR) if as_quat else dR), t0_deform

    @torch.no_grad()
    def _edit(self, x, t, out, node_attrs, node_trans_bias, motion_mask):
        """The Gaussians re-posed around dragged nodes: every Gaussian keeps its offset to its (graph-)nearest nodes, rotated
        with them (time_utils.py:1165-1213).  ``out``: the blend without the drag."""
        rot_bias = torch.tensor([1.0, 0.0, 0.0, 0.0], device=x.device)
        node_trans = node_attrs["d_xyz"]
        cur_node = (self.nodes[..., :3] + node_trans).detach()
        nodes_t = cur_node + node_trans_bias
        gs_init = x + out["d_xyz"]
        if not self.d_rot_as_res:
            node_rot = node_attrs["d_rotation"]  # (already bias-multiplied by forward())
            d2, idx = _knn_sq(gs_init, cur_node, 32)
            w = torch.exp(-d2 / (2 * self.node_radius[idx] ** 2))
            if self.with_node_weight:
                w = w * self.node_weight[idx][..., 0]
            w = w + 1e-7
            w = w / w.sum(dim=-1, keepdim=True)
            R = _quat_to_mat(node_rot + rot_bias)[idx]
            gs_t = nodes_t[idx] + torch.einsum("gkab,gkb->gka", R, gs_init[:, None] - cur_node[idx])
            out["d_xyz"] = ((gs_t * w[..., None]).sum(dim=1) - x) * motion_mask
            return out
        w, _, idx = self.cal_nn_weight_floyd(x=gs_init, t0=t, cur_node=cur_node, K=8, GraphK=3, temperature=1e-3, XisNode=False)
        q_bias, _ = self.p2dR(p=nodes_t, p0=cur_node, K=8, as_quat=True, mode="trajectory", t0=t)
        R = _quat_to_mat(q_bias)[idx]
        gs_t = nodes_t[idx] + torch.einsum("gkab,gkb->gka", R, gs_init[:, None] - cur_node[idx])
        out["d_xyz"] = ((gs_t * w[..., None]).sum(dim=1) - x) * motion_mask
        out["d_rotation_bias"] = ((q_bias[idx] * w[..., None]).sum(dim=1) - rot_bias) * motion_mask + rot_bias
        return out

    def forward(self, x, t, feature, motion_mask, animation_d_values=None, node_trans_bias=None, **kwargs):
        if t.dim() == 0:
            t = self.expand_time(t)
        node_attrs = dict(self.node_deform(t=t))
        if animation_d_values is not None:
            node_attrs.update(animation_d_values)
        if node_trans_bias is not None:
            if self.skinning:
                raise NotImplementedError("node_trans_bias with skinning: the reference's own forward fails there (cal_nn_weight "
                                          "is asked for the weights of feature=None, time_utils.py:936)")
            if not self.d_rot_as_res:  # the dragged nodes' rotations enter the blend itself (time_utils.py:1165-1172)
                with torch.no_grad():
                    rb = torch.tensor([1.0, 0.0, 0.0, 0.0], device=x.device)
                    cur = (self.nodes[..., :3] + node_attrs["d_xyz"]).detach()
                    q_bias, _ = self.p2dR(p=cur + node_trans_bias, p0=cur, K=8, as_quat=True, mode="trajectory", t0=t)
                node_attrs["d_rotation"] = _quat_mul(q_bias, node_attrs["d_rotation"] + rb) - rb
        if self.skinning:
            out = skinning_blend(feature, motion_mask, node_attrs, self.d_rot_as_res, self.pred_opacity, self.pred_color)
        else:
            nw = self._node_weight if self.with_node_weight else None
            out = control_node_blend(x, feature, motion_mask, self.nodes, self._node_radius, nw, node_attrs, K=self.K,
                                     hyper_dim=self.hyper_dim, local_frame=self.local_frame, d_rot_as_res=self.d_rot_as_res)
            if self.pred_opacity or self.pred_color:  # (time_utils.py:1214-1225) the same weights, seen by autograd
                w = knn_weights_torch(x, feature, self.nodes, self._node_radius, nw, out["nn_idx"], self.hyper_dim)
                idx = out["nn_idx"].long()
                if self.pred_opacity:
                    out["d_opacity"] = (node_attrs["d_opacity"][idx] * w[..., None]).sum(dim=1) * motion_mask
                if self.pred_color:
                    out["d_color"] = (node_attrs["d_color"][idx] * w[..., None]).sum(dim=1) * motion_mask
        out["d_nodes"] = self.nodes[..., :3] + node_attrs["d_xyz"]
        if node_trans_bias is not None:
            out = self._edit(x.detach(), t, out, node_attrs, node_trans_bias, motion_mask)
        return out
