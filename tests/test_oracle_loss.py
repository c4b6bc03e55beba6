"""The image-loss oracle against values and autograd gradients of the reference's own l1_loss / ssim."""
import os

import numpy as np

from oracle import loss_ref as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_l1_ssim.npz"))


def test_values_match_reference():
    assert abs(O.l1(G["image"], G["gt"]) - float(G["l1"])) < 1e-7
    assert abs(O.ssim(G["image"], G["gt"]) - float(G["ssim"])) < 1e-6
    lam = float(G["lambda_dssim"])
    assert abs((1 - lam) * O.l1(G["image"], G["gt"]) + lam * (1 - O.ssim(G["image"], G["gt"])) - float(G["loss"])) < 1e-6


def test_gradients_match_reference_autograd():
    lam = float(G["lambda_dssim"])
    for key, (a, b) in {"grad_l1": (1.0, 0.0), "grad_ssim": (0.0, 1.0), "grad_loss": (1 - lam, -lam)}.items():
        g = O.grad(G["image"], G["gt"], a, b)
        assert np.abs(g - G[key]).max() <= 2e-5 * np.abs(G[key]).max(), key
    # exact zero of the L1 gradient where image == gt (torch.abs backward), flat black region handled
    assert (G["grad_l1"][:, -3:, :] == 0).all() and (O.grad(G["image"], G["gt"], 1.0, 0.0)[:, -3:, :] == 0).all()


# ---- skeleton projection loss: the reference's own sampling / projection / composition (chamfer factor restated) ----------
SKEL = ["skelproj_tree24_m700", "skelproj_chain8_m90_K"]


def _skel(name):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    intr = O.intrinsics(float(z["FoVx"]), float(z["FoVy"]), int(z["image_height"]), int(z["image_width"]), z["K"])
    return z, intr


def test_skeleton_sampling_and_projection_match_reference():
    for name in SKEL:
        z, intr = _skel(name)
        t = O.sampling_steps(z["d_nodes"], z["parents"])
        assert len(t) == int(z["steps"])
        pts = O.sampling_skeleton_points(z["d_nodes"], z["parents"], t)
        assert np.abs(pts - z["sampling_points"]).max() < 2e-7
        proj = O.project_nodes(z["sampling_points"], z["world_view_transform"], *intr)
        assert np.abs(proj - z["projected"]).max() < 5e-5  # float32 pixels of O(100) in the fixture


def test_skeleton_projection_loss_and_gradient_match_reference_autograd():
    for name in SKEL:
        z, intr = _skel(name)
        loss, g = O.skeleton_projection_loss(z["d_nodes"], z["parents"], z["world_view_transform"], *intr, z["thinned"])
        assert abs(loss - float(z["loss"])) < 1e-5 * float(z["loss"])
        assert np.abs(g - z["grad_nodes"]).max() <= 1e-5 * np.abs(z["grad_nodes"]).max()


def test_skeleton_projection_gradient_is_the_derivative_of_the_loss():
    z, intr = _skel(SKEL[1])
    t = O.sampling_steps(z["d_nodes"], z["parents"])
    nodes = z["d_nodes"].astype(np.float64)
    _, g = O.skeleton_projection_loss(nodes, z["parents"], z["world_view_transform"], *intr, z["thinned"], t=t)
    rng = np.random.default_rng(0)
    d = rng.standard_normal(nodes.shape)
    h = 1e-7  # small enough that no nearest neighbour changes (piecewise-linear loss)
    lp, _ = O.skeleton_projection_loss(nodes + h * d, z["parents"], z["world_view_transform"], *intr, z["thinned"], t=t)
    lm, _ = O.skeleton_projection_loss(nodes - h * d, z["parents"], z["world_view_transform"], *intr, z["thinned"], t=t)
    assert abs((lp - lm) / (2 * h) - (g * d).sum()) < 1e-3 * abs((g * d).sum())


# ---- the stage-2 objective's regularisers against the reference's own render_and_cal_loss (both cameras) ------------------------
OBJ = np.load(os.path.join(os.path.dirname(__file__), "golden", "objective_tree8_n48.npz"))


def test_stage2_regularisers_match_the_reference_objective():
    for tag in ("template", "other"):
        is_t = int(OBJ[tag + "_uid"]) == int(OBJ["template_idx"])
        r = O.stage2_regularisers(OBJ["template_offsets"], OBJ["local_rotation"], is_t, float(OBJ["lambda_template_offsets"]),
                                  float(OBJ["lambda_template_fixed"]))
        assert abs(r["template_offsets_loss"] - float(OBJ[tag + "_template_offsets_loss"])) <= 1e-6 * r["template_offsets_loss"]
        assert r["lambda_template_offsets"] == float(OBJ[tag + "_lambda_template_offsets"])
        g = OBJ[tag + "_g_template_offsets"]
        assert np.abs(r["g_template_offsets"] - g).max() <= 1e-6 * np.abs(g).max()
        gq = OBJ[tag + "_g_local_rotation"]
        if is_t:
            assert abs(r["template_fixed_loss"] - float(OBJ[tag + "_template_fixed_loss"])) <= 1e-6 * r["template_fixed_loss"]
            assert np.abs(r["g_local_rotation"] - gq).max() <= 1e-6 * np.abs(gq).max()
        else:
            assert r["template_fixed_loss"] is None and not gq.any() and not r["g_local_rotation"].any()
        # the total: the image term (the reference logs it) + the regularisers
        total = float(OBJ["lambda_rendering_image"]) * float(OBJ[tag + "_loss_img"]) + r["total"]
        assert abs(total - float(OBJ[tag + "_loss"])) <= 2e-6 * abs(total)
    # the image term of the golden is the reference's l1 / ssim of the stand-in render: the image-loss oracle reproduces it
    img, gt = OBJ["template_render"].astype(np.float64), OBJ["gt_image"].astype(np.float64)
    lam = float(OBJ["lambda_dssim"])
    assert abs((1 - lam) * O.l1(img, gt) + lam * (1 - O.ssim(img, gt)) - float(OBJ["template_loss_img"])) < 2e-6
