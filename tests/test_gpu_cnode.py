"""§8-f rank 4 (second half) on the GPU: the stage-1 control-node deformation against the reference's goldens and the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ["cnodes_local_res_h8", "cnodes_global_abs_h0", "cnodes_default_h8"]
ATTRS = ("d_xyz", "d_rotation", "d_scaling", "local_rotation")
OUTS = ("d_xyz", "d_rotation", "d_scaling", "d_nodes")


def _module(z):
    from riggs_amd.control_nodes import ControlNodeWarp
    cn = ControlNodeWarp(node_num=z["nodes"].shape[0], K=int(z["K"]), with_node_weight=bool(z["with_node_weight"]),
                         local_frame=bool(z["local_frame"]), d_rot_as_res=bool(z["d_rot_as_res"]), hyper_dim=int(z["hyper_dim"])).cuda()
    cn.nodes.data = torch.from_numpy(z["nodes"]).cuda()
    cn._node_radius.data = torch.from_numpy(z["_node_radius"]).cuda()
    if bool(z["with_node_weight"]):
        cn._node_weight.data = torch.from_numpy(z["_node_weight"]).cuda()
    return cn


@pytest.mark.parametrize("name", NAMES)
def test_module_matches_reference_golden(name):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    cn = _module(z)
    x = torch.from_numpy(z["x"]).cuda()
    feature = torch.from_numpy(z["feature"]).cuda().requires_grad_(True) if z["feature"].size else None
    mask = torch.from_numpy(z["motion_mask"]).cuda().requires_grad_(True)
    attrs = {k: torch.from_numpy(z["attr_" + k]).cuda().requires_grad_(True) for k in ATTRS}
    out = cn(x, torch.tensor(0.3, device="cuda"), feature, mask, animation_d_values=attrs)
    assert torch.equal(out["nn_idx"].cpu(), torch.from_numpy(z["nn_idx"]))          # index work: exact
    assert np.abs(out["nn_weight"].cpu().numpy() - z["nn_weight"]).max() < 2e-6
    for k in OUTS:
        assert np.abs(out[k].detach().cpu().numpy() - z["out_" + k]).max() <= 1e-5 * max(1.0, np.abs(z["out_" + k]).max()), k
    sum((out[k] * torch.from_numpy(z["gout_" + k]).cuda()).sum() for k in OUTS).backward()
    got = {"attr_" + k: v.grad for k, v in attrs.items()}
    got.update(nodes=cn.nodes.grad, _node_radius=cn._node_radius.grad, motion_mask=mask.grad)
    if feature is not None:
        got["feature"] = feature.grad
    if bool(z["with_node_weight"]):
        got["_node_weight"] = cn._node_weight.grad
    for k, g in got.items():
        ref = z["grad_" + k]
        if g is None:
            assert np.abs(ref).max() == 0, k
            continue
        err = np.abs(g.cpu().numpy().reshape(ref.shape) - ref).max()
        assert err <= 1e-4 * max(np.abs(ref).max(), 1.0), (k, err)


@pytest.mark.parametrize("N,M,K,hyper,local,res,nw,use_mask", [
    (20_000, 512, 3, 8, True, True, True, True),      # the shipped shape of stage 1
    (5_000, 1024, 3, 8, False, True, True, False),    # arguments/__init__.py defaults (node_num 1024, local_frame False)
    (3_001, 40, 8, 0, True, False, False, True),      # K = 8, xyz only, absolute rotation
    (257, 5, 5, 2, False, False, True, False),        # K = M
    (1, 64, 3, 11, True, True, False, True),          # one Gaussian, widest hyper
])
def test_against_oracle(N, M, K, hyper, local, res, nw, use_mask):
    from oracle import cnode_ref as O
    from riggs_amd.control_nodes import control_node_blend
    g = torch.Generator().manual_seed(N + M)
    x = torch.randn(N, 3, generator=g) * 0.5
    nodes = torch.cat([x[torch.randint(0, N, (M,), generator=g)] + 0.05 * torch.randn(M, 3, generator=g),
                       1e-2 + 0.02 * torch.randn(M, hyper, generator=g)], -1)
    feature = 0.02 * torch.randn(N, hyper + 1, generator=g) if hyper else None
    mask = torch.rand(N, 1, generator=g) if use_mask else None
    radius = np.log(0.15) + 0.3 * torch.randn(M, generator=g)
    weight = 0.5 * torch.randn(M, 1, generator=g) if nw else None
    attrs = {"d_xyz": 0.1 * torch.randn(M, 3, generator=g), "d_rotation": 0.2 * torch.randn(M, 4, generator=g),
             "d_scaling": 0.05 * torch.randn(M, 3, generator=g), "local_rotation": 0.3 * torch.randn(M, 4, generator=g)}
    gout = {k: torch.randn(N, w, generator=g) for k, w in (("d_xyz", 3), ("d_rotation", 4), ("d_scaling", 3))}
    cu = lambda t: None if t is None else t.cuda().requires_grad_(True)  # noqa: E731
    c_feat, c_mask, c_nodes, c_rad, c_w = cu(feature), cu(mask), cu(nodes), cu(radius), cu(weight)
    c_attrs = {k: cu(v) for k, v in attrs.items()}
    out = control_node_blend(x.cuda(), c_feat, c_mask, c_nodes, c_rad, c_w, c_attrs, K=K, hyper_dim=hyper, local_frame=local,
                             d_rot_as_res=res)
    sum((out[k] * gout[k].cuda()).sum() for k in gout).backward()
    npy = lambda t: None if t is None else t.numpy()  # noqa: E731
    cfg = dict(K=K, hyper_dim=hyper, local_frame=local, d_rot_as_res=res)
    m_np = npy(mask) if mask is not None else np.ones((N, 1), np.float32)
    a_np = {k: v.numpy() for k, v in attrs.items()}
    ref = O.forward(x.numpy(), npy(feature), m_np, nodes.numpy(), radius.numpy(), npy(weight), a_np, **cfg)
    idx = out["nn_idx"].cpu().numpy()
    same = (idx == ref["nn_idx"]).all(1)
    assert same.mean() > 0.999  # float32 vs float64 near-ties of the K-th neighbour may swap; everything else: exact
    for k in gout:
        err = np.abs(out[k].detach().cpu().numpy() - ref[k])[same].max()
        assert err <= 1e-5 * max(1.0, np.abs(ref[k]).max()), (k, err)
    if not same.all():
        return  # gradients are sums over Gaussians: only comparable when every neighbour list agrees
    go = {k: v.numpy() for k, v in gout.items()}
    go["d_nodes"] = np.zeros((M, 3), np.float32)
    gref = O.backward(x.numpy(), npy(feature), m_np, nodes.numpy(), radius.numpy(), npy(weight), a_np, gout=go, **cfg)
    got = dict(c_attrs)
    got.update(nodes=c_nodes, _node_radius=c_rad)
    if weight is not None:
        got["_node_weight"] = c_w
    if feature is not None:
        got["feature"] = c_feat
    if mask is not None:
        got["motion_mask"] = c_mask
    for k, t in got.items():
        if k == "local_rotation" and not local:
            assert t.grad is None or float(t.grad.abs().max()) == 0
            continue
        r = gref[k]
        err = np.abs(t.grad.cpu().numpy().reshape(r.shape) - r).max()
        assert err <= 2e-4 * max(np.abs(r).max(), 1.0), (k, err, np.abs(r).max())


def test_no_feature_means_xyz_only_and_unsupported_options_raise():
    from riggs_amd import _lib as L
    from riggs_amd.control_nodes import ControlNodeWarp, control_node_blend
    cn = ControlNodeWarp(node_num=32, K=3, hyper_dim=4).cuda()
    x = torch.randn(100, 3, device="cuda")
    out = cn(x, torch.tensor(0.1, device="cuda"), None, 1.0)
    d = ((x[:, None] - cn.nodes[None, :, :3]) ** 2).sum(-1)
    assert torch.equal(out["nn_idx"].long(), d.topk(3, dim=1, largest=False).indices)
    assert float(out["d_xyz"].detach().abs().max()) == 0 and out["d_nodes"].shape == (32, 3)   # static network: zero deformation
    with pytest.raises(NotImplementedError):
        ControlNodeWarp(skinning=True, local_frame=True)  # (the reference's own forward fails in this combination)
    with pytest.raises(NotImplementedError):  # (the reference's own forward fails here as well)
        ControlNodeWarp(node_num=8, skinning=True).cuda()(x, torch.tensor(0.1, device="cuda"), torch.randn(100, 8, device="cuda"), 1.0,
                                                          node_trans_bias=torch.zeros(8, 3, device="cuda"))
    with pytest.raises(L.RiggsHipError):
        control_node_blend(x.cpu(), None, None, cn.nodes, cn._node_radius, None, cn.node_deform(torch.zeros(32, 1, device="cuda")))
    with pytest.raises(L.RiggsHipError):  # K > 8
        control_node_blend(x, None, None, cn.nodes, cn._node_radius, None, cn.node_deform(torch.zeros(32, 1, device="cuda")), K=9)


class _NodeNet(torch.nn.Module):
    """A small stand-in for the reference's DeformNetwork (nodes, t -> per-node attributes); stage-1's node network is a
    torch module supplied by the caller."""

    def __init__(self):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Linear(4, 32), torch.nn.ReLU(), torch.nn.Linear(32, 14))

    def forward(self, x, t, **kw):
        o = self.body(torch.cat([x, t], -1)) * torch.tensor([.05] * 3 + [.1] * 4 + [.02] * 3 + [.2] * 4, device=x.device)
        return {"d_xyz": o[:, :3], "d_rotation": o[:, 3:7], "d_scaling": o[:, 7:10], "local_rotation": o[:, 10:14],
                "hidden": None, "d_opacity": None, "d_color": None}


def _torch_blend(cn, x, feature, mask, attrs):
    """ControlNodeWarp.forward's arithmetic in torch ops (time_utils.py:934-964, 1138-1191), differentiable by autograd."""
    h = cn.hyper_dim
    xa = torch.cat([x, feature[:, :h]], -1)
    na = torch.cat([cn.nodes[:, :3].detach(), cn.nodes[:, 3:]], -1)
    d, idx = ((xa[:, None] - na[None]) ** 2).sum(-1).topk(cn.K, dim=1, largest=False)
    w = torch.exp(-d / (2 * cn.node_radius[idx] ** 2)) * cn.node_weight[idx][..., 0] + 1e-7
    w = w / w.sum(-1, keepdim=True)
    q = attrs["local_rotation"] + torch.tensor([1.0, 0, 0, 0], device=x.device)
    r, i, j, k = torch.unbind(q, -1)
    s = 2.0 / (q * q).sum(-1)
    R = torch.stack((1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r), s * (i * j + k * r),
                     1 - s * (i * i + k * k), s * (j * k - i * r), s * (i * k - j * r), s * (j * k + i * r),
                     1 - s * (i * i + j * j)), -1).reshape(-1, 3, 3)
    nn_ = cn.nodes[idx, :3].detach()
    Ax = torch.einsum("nkab,nkb->nka", R[idx], x[:, None] - nn_) + nn_ + attrs["d_xyz"][idx]
    return {"d_xyz": ((Ax * w[..., None]).sum(1) - x) * mask, "d_rotation": (attrs["d_rotation"][idx] * w[..., None]).sum(1) * mask,
            "d_scaling": (attrs["d_scaling"][idx] * w[..., None]).sum(1) * mask}


def test_stage1_iteration_through_the_rasterizer_matches_torch_blend():
    """deform (control nodes) -> render -> image loss -> backward: the gradients that reach the node network, the nodes, the
    radii / weights and the Gaussians' hyper feature agree with the same chain whose blend is written in torch ops."""
    from riggs_amd import synth
    from riggs_amd.control_nodes import ControlNodeWarp
    from riggs_amd.gaussian_model import GaussianModel
    from riggs_amd.graph import _Pipe
    from riggs_amd.loss import image_loss
    from riggs_amd.render import render
    N, M, H = 6_000, 96, 8
    sc = synth.make_scene(N, 8, 7)
    cam = synth.look_at_camera(96, 96, fid=0.4).to("cuda")
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                    sc["opacity"], device="cuda")
    gm.fea_dim, gm.with_motion_mask = H + 1, True
    torch.manual_seed(5)
    gm.feature = torch.nn.Parameter(torch.cat([0.02 * torch.randn(N, H), 2.0 + torch.randn(N, 1)], -1).cuda())
    cn = ControlNodeWarp(node_num=M, K=3, local_frame=True, d_rot_as_res=True, hyper_dim=H, network=_NodeNet()).cuda()
    with torch.no_grad():
        cn.nodes.copy_(torch.cat([gm.get_xyz[torch.randperm(N)[:M].cuda()], 1e-2 + 0.02 * torch.randn(M, H, device="cuda")], -1))
        cn._node_radius.fill_(float(np.log(0.2)))
        cn._node_weight.copy_(0.3 * torch.randn(M, 1))
    target = torch.rand(3, 96, 96, device="cuda")
    bg = torch.zeros(3, device="cuda")
    leaves = {"net": list(cn.network.parameters())[0], "nodes": cn.nodes, "radius": cn._node_radius, "weight": cn._node_weight,
              "feature": gm.feature, "xyz": gm._xyz}
    res = {}
    for which in ("hip", "torch"):
        for t in list(leaves.values()) + list(cn.network.parameters()):
            t.grad = None
        tt = torch.tensor(0.4, device="cuda")
        if which == "hip":
            dv = cn(gm.get_xyz.detach(), tt, gm.feature, gm.motion_mask)
        else:
            dv = _torch_blend(cn, gm.get_xyz.detach(), gm.feature, gm.motion_mask, cn.node_deform(cn.expand_time(tt)))
        pkg = render(cam, gm, _Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"], d_rot_as_res=True)
        loss, _ = image_loss(pkg["render"], target, 0.2)
        loss.backward()
        res[which] = (loss.item(), {k: v.grad.clone() for k, v in leaves.items()})
    assert abs(res["hip"][0] - res["torch"][0]) <= 1e-5 * res["torch"][0]
    for k, g in res["torch"][1].items():
        err = (res["hip"][1][k] - g).abs().max().item()
        assert g.abs().max().item() > 0 and err <= 2e-3 * g.abs().max().item(), (k, err, g.abs().max().item())


def test_atomics_backward_variant_passes_the_same_goldens():
    """riggs_set_option("cnode_bwd_atomics", 1) selects the first backward (LDS float atomics + partial tables): the golden
    and oracle cases once more with it (workspaces are sized per call, after the option is set)."""
    from riggs_amd import _lib as L
    assert L.get_option("cnode_bwd_atomics") == 0
    L.set_option("cnode_bwd_atomics", 1)
    try:
        for name in NAMES:
            test_module_matches_reference_golden(name)
        test_against_oracle(20_000, 512, 3, 8, True, True, True, True)
        test_against_oracle(3_001, 40, 8, 0, True, False, False, True)
        test_against_oracle(1, 64, 3, 11, True, True, False, True)
    finally:
        L.set_option("cnode_bwd_atomics", 0)


def test_pred_opacity_and_color_match_reference_golden():
    """KNN weights (HIP kernel) + the opacity / colour blends of time_utils.py:1214-1225, whose gradients reach the node radii,
    node weights, hyper coordinates and the Gaussians' feature through the weights themselves."""
    from riggs_amd.control_nodes import ControlNodeWarp
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnodes_knn_pred_opacity_color.npz"))
    T = lambda k, rg=False: torch.from_numpy(g[k].copy()).cuda().requires_grad_(rg)  # noqa: E731
    M = g["nodes"].shape[0]
    cn = ControlNodeWarp(node_num=M, K=3, hyper_dim=2, pred_opacity=True, pred_color=True).cuda()
    cn.nodes.data, cn._node_radius.data, cn._node_weight.data = T("nodes"), T("_node_radius"), T("_node_weight")
    feature, mask = T("feature", True), T("motion_mask", True)
    attrs = {k: T("attr_" + k, True) for k in ("d_xyz", "d_rotation", "d_scaling", "local_rotation", "d_opacity", "d_color")}
    out = cn(T("x"), torch.tensor(0.3).cuda(), feature, mask, animation_d_values=attrs)
    keys = ("d_xyz", "d_rotation", "d_scaling", "d_opacity", "d_color")
    for k in keys:
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), g["out_" + k], rtol=2e-5, atol=2e-6)
    sum((out[k] * T("gout_" + k)).sum() for k in keys).backward()
    rel = lambda a, b: float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))  # noqa: E731
    assert rel(feature.grad.cpu().numpy(), g["grad_feature"]) < 2e-4
    assert rel(mask.grad.cpu().numpy(), g["grad_motion_mask"]) < 2e-4
    assert rel(cn._node_radius.grad.cpu().numpy(), g["grad__node_radius"]) < 2e-4
    assert rel(cn._node_weight.grad.cpu().numpy(), g["grad__node_weight"]) < 2e-4
    assert rel(cn.nodes.grad.cpu().numpy(), g["grad_nodes"]) < 2e-4
    for k in ("d_xyz", "d_rotation", "d_scaling", "d_opacity", "d_color"):
        assert rel(attrs[k].grad.cpu().numpy(), g["grad_attr_" + k]) < 2e-4, k


class _WavingNodes(torch.nn.Module):
    """The closed-form node network of the editing fixtures (tests/golden/make_golden.py: WavingNodes)."""

    def forward(self, x, t, **kwargs):
        z3, z4 = torch.zeros_like(x), torch.zeros(x.shape[0], 4, dtype=x.dtype, device=x.device)
        return {"d_xyz": 0.08 * torch.sin(6.283185307179586 * t + 3.0 * x), "d_rotation": z4, "d_scaling": z3,
                "local_rotation": z4.clone(), "hidden": None, "d_opacity": None, "d_color": None}


@pytest.mark.parametrize("name", ["cnodes_edit_res_m96", "cnodes_edit_abs_m64"])
def test_node_trans_bias_editing_path_matches_reference_golden(name):
    """ControlNodeWarp.forward(node_trans_bias=...) — the GUI's drag-to-edit path (utils/time_utils.py:1165-1213: Kabsch rotations
    of the dragged nodes from their trajectory neighbours, graph-distance weights, the Gaussians re-posed rigidly around their
    nodes) — against outputs of the reference module itself (generated by tests/golden/make_golden.py)."""
    from riggs_amd.control_nodes import ControlNodeWarp
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    cu = lambda a: torch.from_numpy(np.asarray(a)).cuda()  # noqa: E731
    cn = ControlNodeWarp(node_num=z["nodes"].shape[0], K=3, with_node_weight=True, local_frame=False, d_rot_as_res=bool(z["d_rot_as_res"]),
                         hyper_dim=2, network=_WavingNodes()).cuda()
    with torch.no_grad():
        cn.nodes.copy_(cu(z["nodes"])); cn._node_radius.copy_(cu(z["_node_radius"])); cn._node_weight.copy_(cu(z["_node_weight"]))
    with torch.no_grad():
        out = cn(cu(z["x"]), torch.tensor(float(z["t"]), device="cuda"), cu(z["feature"]), cu(z["motion_mask"]),
                 node_trans_bias=cu(z["node_trans_bias"]))
    keys = ["d_xyz", "d_rotation", "d_scaling"] + (["d_rotation_bias"] if "out_d_rotation_bias" in z.files else [])
    for k in keys:
        ref = z["out_" + k]
        err = np.abs(out[k].cpu().numpy() - ref).max()
        assert err <= 2e-4 * max(1.0, np.abs(ref).max()), (k, err)
    # the drag moved something (a zero drag is NOT the plain blend in the reference either: the re-posing uses other weights)
    with torch.no_grad():
        plain = cn(cu(z["x"]), torch.tensor(float(z["t"]), device="cuda"), cu(z["feature"]), cu(z["motion_mask"]))
    assert float((out["d_xyz"] - plain["d_xyz"]).abs().max()) > 1e-2
