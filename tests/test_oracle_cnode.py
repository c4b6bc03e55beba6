"""The control-node oracle against outputs and autograd gradients of the reference's own ControlNodeWarp (CPU goldens)."""
import os

import numpy as np
import pytest

from oracle import cnode_ref as O

NAMES = ["cnodes_local_res_h8", "cnodes_global_abs_h0", "cnodes_default_h8"]
ATTRS = ("d_xyz", "d_rotation", "d_scaling", "local_rotation")


def load(name):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    feat = z["feature"] if z["feature"].size else None
    nw = z["_node_weight"] if bool(z["with_node_weight"]) else None
    attrs = {k: z["attr_" + k] for k in ATTRS}
    cfg = dict(K=int(z["K"]), hyper_dim=int(z["hyper_dim"]), local_frame=bool(z["local_frame"]), d_rot_as_res=bool(z["d_rot_as_res"]))
    return z, feat, nw, attrs, cfg


@pytest.mark.parametrize("name", NAMES)
def test_forward_matches_reference(name):
    z, feat, nw, attrs, cfg = load(name)
    o = O.forward(z["x"], feat, z["motion_mask"], z["nodes"], z["_node_radius"], nw, attrs, **cfg)
    assert (o["nn_idx"] == z["nn_idx"]).all()
    assert np.abs(o["nn_weight"] - z["nn_weight"]).max() < 2e-6 and np.abs(o["nn_dist"] - z["nn_dist"]).max() < 2e-6
    for k in ("d_xyz", "d_rotation", "d_scaling", "d_nodes"):
        assert np.abs(o[k] - z["out_" + k]).max() < 2e-6, k


@pytest.mark.parametrize("name", NAMES)
def test_gradients_match_reference_autograd(name):
    z, feat, nw, attrs, cfg = load(name)
    gout = {k: z["gout_" + k] for k in ("d_xyz", "d_rotation", "d_scaling", "d_nodes")}
    g = O.backward(z["x"], feat, z["motion_mask"], z["nodes"], z["_node_radius"], nw, attrs, gout=gout, **cfg)
    for k, v in g.items():
        ref = z[("grad_attr_" if k in ATTRS else "grad_") + k]
        assert np.abs(v - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1.0), k


@pytest.mark.parametrize("name", ["cnodes_skinning_m48", "cnodes_skinning_abs_m32"])
def test_skinning_mode_matches_reference_golden(name):
    """ControlNodeWarp(skinning=True) — a torch-op path (dense (N, M) products), so it is checked here on the CPU against the
    reference module's own outputs and autograd gradients (tests/golden/make_golden.py:fixture_control_nodes_blends)."""
    import torch
    from riggs_amd.control_nodes import ControlNodeWarp
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    T = lambda k, rg=False: torch.from_numpy(g[k].copy()).requires_grad_(rg)  # noqa: E731
    M = g["nodes"].shape[0]
    cn = ControlNodeWarp(node_num=M, K=3, skinning=True, d_rot_as_res=bool(g["d_rot_as_res"]), pred_opacity=True, pred_color=True)
    cn.nodes.data = T("nodes")
    assert [d["name"] for d in cn.trainable_parameters()] == ["deform", "nodes"] and not hasattr(cn, "_node_radius")
    feature, mask = T("feature", True), T("motion_mask", True)
    attrs = {k: T("attr_" + k, True) for k in ("d_xyz", "d_rotation", "d_scaling", "local_rotation", "d_opacity", "d_color")}
    out = cn(T("x"), torch.tensor(0.3), feature, mask, animation_d_values=attrs)
    keys = ("d_xyz", "d_rotation", "d_scaling", "d_opacity", "d_color")
    for k in keys:
        np.testing.assert_allclose(out[k].detach().numpy(), g["out_" + k], rtol=1e-5, atol=1e-6)
    sum((out[k] * T("gout_" + k)).sum() for k in keys).backward()
    np.testing.assert_allclose(feature.grad.numpy(), g["grad_feature"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(mask.grad.numpy(), g["grad_motion_mask"], rtol=1e-4, atol=1e-6)
    for k in ("d_xyz", "d_rotation", "d_scaling", "d_opacity", "d_color"):
        np.testing.assert_allclose(attrs[k].grad.numpy(), g["grad_attr_" + k], rtol=1e-4, atol=1e-6)
