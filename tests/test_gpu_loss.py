"""§8-f rank 2 on the GPU: fused L1 + SSIM forward / backward against the reference's golden vectors and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_l1_ssim.npz"))


def test_trainer_usage_matches_reference_golden():
    from riggs_amd.loss import l1_loss, ssim
    lam = float(G["lambda_dssim"])
    gt = torch.from_numpy(G["gt"]).cuda()
    x = torch.from_numpy(G["image"]).cuda().requires_grad_(True)
    Ll1 = l1_loss(x, gt)                       # train_rig.py:508
    s = ssim(x, gt)                            # :509 — same fused node
    assert Ll1.grad_fn is s.grad_fn
    loss = (1.0 - lam) * Ll1 + lam * (1.0 - s)
    loss.backward()
    assert abs(Ll1.item() - float(G["l1"])) < 1e-6 and abs(s.item() - float(G["ssim"])) < 2e-6
    assert abs(loss.item() - float(G["loss"])) < 2e-6
    g = x.grad.cpu().numpy()
    assert np.abs(g - G["grad_loss"]).max() <= 1e-4 * np.abs(G["grad_loss"]).max()
    for key, pick in (("grad_l1", 0), ("grad_ssim", 1)):
        x = torch.from_numpy(G["image"]).cuda().requires_grad_(True)
        (l1_loss(x, gt), ssim(x, gt))[pick].backward()
        assert np.abs(x.grad.cpu().numpy() - G[key]).max() <= 1e-4 * np.abs(G[key]).max(), key
        if key == "grad_l1":  # image == gt on the last rows: sign(0) = 0, as torch.abs' backward
            assert (x.grad[:, -3:, :] == 0).all()


@pytest.mark.parametrize("C,H,W", [(3, 800, 800), (1, 17, 5), (3, 33, 129), (4, 16, 16)])
def test_against_oracle_ragged_and_full_size(C, H, W):
    from oracle import loss_ref as O
    from riggs_amd.loss import l1_ssim
    g = torch.Generator().manual_seed(C * 1000 + H)
    gt = torch.rand(C, H, W, generator=g)
    img = (gt + 0.1 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
    x = img.cuda().requires_grad_(True)
    l1, s = l1_ssim(x, gt.cuda())
    (0.7 * l1 - 0.3 * s).backward()
    if H * W <= 200 * 200:
        assert abs(l1.item() - O.l1(img.numpy(), gt.numpy())) < 1e-6
        assert abs(s.item() - O.ssim(img.numpy(), gt.numpy())) < 5e-6
        want = O.grad(img.numpy(), gt.numpy(), 0.7, -0.3)
        assert np.abs(x.grad.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max()
    else:
        # full size: properties instead of the O(121 HW) oracle — identical images give ssim = 1, l1 = 0 and a zero
        # gradient; the loss is symmetric in its arguments; determinism
        y = gt.cuda()
        a, b = l1_ssim(y.clone().requires_grad_(True), y)
        assert a.item() == 0.0 and abs(b.item() - 1.0) < 1e-6
        l1b, sb = l1_ssim(gt.cuda(), img.cuda())
        assert abs(l1b.item() - l1.item()) < 1e-7 and abs(sb.item() - s.item()) < 1e-6
        x2 = img.cuda().requires_grad_(True)
        l1c, sc = l1_ssim(x2, gt.cuda())
        (0.7 * l1c - 0.3 * sc).backward()
        assert torch.equal(x2.grad, x.grad) and l1c.item() == l1.item() and sc.item() == s.item()


def test_gradient_reaches_the_rasterizer_input():
    """image -> loss -> backward through the fused loss into a leaf (stands in for the rasterizer's output)."""
    from riggs_amd.loss import l1_loss, ssim
    g = torch.Generator().manual_seed(3)
    leaf = torch.rand(3, 64, 48, generator=g).cuda().requires_grad_(True)
    image = leaf * 0.9 + 0.05
    gt = torch.rand(3, 64, 48, generator=g).cuda()
    loss = 0.8 * l1_loss(image, gt) + 0.2 * (1.0 - ssim(image, gt))
    loss.backward()
    assert leaf.grad is not None and torch.isfinite(leaf.grad).all() and float(leaf.grad.abs().sum()) > 0
    with pytest.raises(NotImplementedError):
        ssim(image, gt, window_size=7)
