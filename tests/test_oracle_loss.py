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
