"""Pin oracle/deform_ref.py (CPU restatement) against golden vectors captured from
the real RigGS reference (tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import deform_ref as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEFORM = sorted(glob.glob(os.path.join(GOLD, "deform_*.npz")))
GLUE = sorted(glob.glob(os.path.join(GOLD, "glue_*.npz")))


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("path", DEFORM, ids=[os.path.basename(p)[:-4] for p in DEFORM])
def test_deform_forward_backward_matches_reference(path):
    g = np.load(path)
    joints, parents = T(g["joints"]), T(g["parents"])
    q = T(g["local_rot"]).clone().requires_grad_(True)
    gt = T(g["global_trans"]).clone().requires_grad_(True)
    rho = T(g["node_radius_log"]).clone().requires_grad_(True)
    mask = T(g["motion_mask"]).clone().requires_grad_(True)
    x = T(g["x"])
    K = int(g["K"])
    # G1: FK
    R = O.quaternion_to_matrix(q)
    np.testing.assert_allclose(R.detach().numpy(), g["R"], rtol=0, atol=1e-6)
    posed, G = O.fk_chain(R, joints, parents)
    np.testing.assert_allclose(G.detach().numpy(), g["transforms"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(posed.detach().numpy(), g["posed"], rtol=0, atol=2e-6)
    nr = O.matrix_to_quaternion(G[:, :3, :3].detach())
    np.testing.assert_allclose(nr.numpy(), g["node_rot"], rtol=0, atol=2e-6)
    # G2: weights
    w, d2, idx = O.skin_weights(x, joints, parents, rho, K)
    assert np.array_equal(idx.numpy(), g["nn_idx"])
    np.testing.assert_allclose(d2.detach().numpy(), g["d2"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(w.detach().numpy(), g["nn_weight"], rtol=1e-5, atol=1e-7)
    # G3: LBS forward
    out = O.deform_by_pose(x, joints, parents, rho, q, gt, mask, K)
    for k in ("d_xyz", "d_rotation", "d_scaling", "d_nodes"):
        np.testing.assert_allclose(out[k].detach().numpy(), g[k], rtol=0, atol=2e-6, err_msg=k)
    # G4: backward with the fixture's cotangents
    loss = (out["d_xyz"] * T(g["g_xyz"])).sum() + (out["d_rotation"] * T(g["g_rot"])).sum() \
        + (out["d_nodes"] * T(g["g_nodes"])).sum()
    loss.backward()
    for got, key in ((q.grad, "grad_local_rot"), (gt.grad, "grad_global_trans"), (rho.grad, "grad_node_radius"),
                     (mask.grad, "grad_motion_mask")):
        ref = g[key]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(got.numpy(), ref, rtol=1e-4, atol=2e-5 * scale, err_msg=key)


@pytest.mark.parametrize("path", GLUE, ids=[os.path.basename(p)[:-4] for p in GLUE])
def test_render_glue_and_camera_match_reference(path):
    g = np.load(path)
    m3, op, sc, rot, shs = O.render_glue(T(g["xyz"]), T(g["features_dc"]), T(g["features_rest"]), T(g["scaling"]),
                                         T(g["rotation"]), T(g["opacity"]), T(g["d_xyz"]), T(g["d_rotation"]),
                                         T(g["d_scaling"]), isotropic=bool(g["isotropic"]))
    np.testing.assert_array_equal(m3.numpy(), g["means3D"])
    np.testing.assert_allclose(op.numpy(), g["opacities"], rtol=1e-6)
    np.testing.assert_allclose(sc.numpy(), g["scales"], rtol=1e-6)
    np.testing.assert_allclose(rot.numpy(), g["rotations"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(shs.numpy(), g["shs"])
    K = g["K"] if g["K"].size else None
    wv, full, center = O.camera_matrices(g["cam_R"], g["cam_T"], float(g["fovx"]), float(g["fovy"]), K=K,
                                         W=int(g["W"]), H=int(g["H"]))
    np.testing.assert_allclose(wv.numpy(), g["viewmatrix"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(full.numpy(), g["projmatrix"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(center.numpy(), g["campos"], rtol=0, atol=1e-5)


def test_pose_embedding_matches_reference_posemlp():
    g = np.load(os.path.join(GOLD, "posemlp_w32_j24.npz"))
    from riggs_amd.skeleton import PoseMLP
    J = int(g["J"])
    net = PoseMLP(1, J * 4, depth=8, hidden_dimensions=32, multires=8)
    sd = {k.replace("__", "."): T(g[k]) for k in g.files if "__" in k}
    net.load_state_dict(sd)
    out = net(T(g["t"]))
    np.testing.assert_allclose(out["rotation"].detach().numpy(), g["rotation"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["translation"].detach().numpy(), g["translation"], rtol=1e-5, atol=1e-6)
