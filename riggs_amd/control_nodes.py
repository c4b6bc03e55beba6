"""Stage-1 control-node deformation with the reference's names (/root/reference/utils/time_utils.py:770-1236
``ControlNodeWarp``): K nearest control nodes per Gaussian in (xyz, hyper) space, Gaussian-kernel weights, blend of the nodes'
translations / local-frame rotations / rotation and scale residuals — ONE HIP launch forward, two backward (csrc/cnode.hip)
instead of a KNN extension call, ~10 (N, K, ·) gathers, an einsum and autograd's replay of them.

Covered: the configuration the trainer ships (KNN weights; ``local_frame``, ``d_rot_as_res``, ``with_node_weight``, ``hyper_dim``
free) through the HIP kernels; ``pred_opacity`` / ``pred_color`` (two more blends with the same weights, torch ops on the
kernel's neighbour lists) and ``skinning=True`` (softmax of a per-Gaussian (N, M) feature instead of KNN weights: a dense
(N, M) x (M, 14) product, left to the GEMM library) as the reference defines them; ``node_trans_bias`` (the GUI's editing
path) raises.  The node network
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
        return self.network(x=self.nodes[..., :3].detach(), t=t, **kwargs)

    def forward(self, x, t, feature, motion_mask, animation_d_values=None, node_trans_bias=None, **kwargs):
        if node_trans_bias is not None:
            raise NotImplementedError("node_trans_bias (the editing path, time_utils.py:1165-1213) is out of scope")
        if t.dim() == 0:
            t = self.expand_time(t)
        node_attrs = dict(self.node_deform(t=t))
        if animation_d_values is not None:
            node_attrs.update(animation_d_values)
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
        return out
