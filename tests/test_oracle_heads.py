"""§8-f rank 3 on CPU: the host mirrors of WeightMLP / DeformMLP reproduce the reference's (seeded) weights and outputs,
and the deformation oracle with the heads' outputs matches the reference's deform_by_pose (values and gradients)."""
import os

import numpy as np
import torch

from oracle import deform_ref as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "heads_tree12_n200.npz"))


def seeded_heads(J, seed):
    """Same construction order and post-scaling as tests/golden/make_golden.py:seeded_heads, with the host mirrors."""
    from riggs_amd.skeleton import DeformMLP, WeightMLP
    torch.manual_seed(seed)
    wm = WeightMLP(input_ch=3, output_ch=J - 1)
    dn = DeformMLP(xyz_input_ch=3, time_input_ch=J * 4, t_multires=-1)
    with torch.no_grad():
        dn.gaussian_warp.weight.mul_(2000.0)
        dn.gaussian_warp.bias.add_(0.01)
    return wm, dn


def test_heads_reproduce_reference_weights_and_outputs():
    J = G["joints"].shape[0]
    wm, dn = seeded_heads(J, int(G["head_seed"]))
    for mod, key in ((wm, "chk_wm"), (dn, "chk_dn")):
        chk = np.array([float(p.detach().double().abs().sum()) for p in mod.parameters()])
        np.testing.assert_allclose(chk, G[key], rtol=1e-12)  # same init code path, same RNG stream: bit-identical weights
    x = torch.from_numpy(G["x"])
    np.testing.assert_allclose(wm(x).detach().numpy(), G["skinning_weight_offsets"], rtol=1e-5, atol=1e-6)
    pose = torch.from_numpy(G["local_rot"]).reshape(-1)[None].expand(x.shape[0], -1)
    np.testing.assert_allclose(dn(x, pose).detach().numpy(), G["template_offsets"], rtol=1e-4, atol=1e-6)
    assert [n for n, _ in wm.named_parameters()][:2] == ["linear.0.weight", "linear.0.bias"]  # checkpoint key names


def test_oracle_with_heads_matches_reference():
    T = torch.from_numpy
    J = G["joints"].shape[0]
    wm, dn = seeded_heads(J, int(G["head_seed"]))
    x = T(G["x"])
    q = T(G["local_rot"]).clone().requires_grad_(True)
    gt = T(G["global_trans"]).clone().requires_grad_(True)
    rho = T(G["node_radius"]).clone().requires_grad_(True)
    pose = q.detach().reshape(-1)[None].expand(x.shape[0], -1)
    o = O.deform_by_pose(x, T(G["joints"]), T(G["parents"]), rho, q, gt, T(G["mask"]), -1,
                         template_offsets=dn(x, pose), weight_offsets=wm(x))
    np.testing.assert_allclose(o["d_xyz"].detach().numpy(), G["d_xyz"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(o["d_rotation"].detach().numpy(), G["d_rotation"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(o["nn_weight"].detach().numpy(), G["nn_weight"], rtol=1e-4, atol=1e-7)
    ((o["d_xyz"] * T(G["c_xyz"])).sum() + (o["d_rotation"] * T(G["c_rot"])).sum()).backward()
    for got, key in ((q.grad, "g_local_rot"), (gt.grad, "g_global_trans"), (rho.grad, "g_node_radius")):
        np.testing.assert_allclose(got.numpy(), G[key], rtol=2e-4, atol=2e-5 * np.abs(G[key]).max())
    sd = {"wm": dict(wm.named_parameters()), "dn": dict(dn.named_parameters())}
    for k in G.files:
        if k.startswith("g_wm_") or k.startswith("g_dn_"):
            got = sd[k[2:4]][k[5:]].grad.numpy()
            np.testing.assert_allclose(got, G[k], rtol=1e-3, atol=2e-5 * np.abs(G[k]).max(), err_msg=k)
