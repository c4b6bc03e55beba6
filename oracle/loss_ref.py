"""CPU restatement of the trainer's image loss and of its gradient.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) — never imported by the product.

* ``l1`` — /root/reference/utils/loss_utils.py:17-18.
* ``ssim`` — loss_utils.py:33-77: 11-tap Gaussian (sigma 1.5, float32, normalised) as a 2-D outer-product window, zero padding
  (``padding=window_size // 2``), per-channel (``groups=channel``), C1 = 0.01^2, C2 = 0.03^2, mean over every element.
* ``grad`` — d(a * l1 + b * ssim)/d(image), derived by hand (the reference leaves it to autograd):
  with the five windowed moments mu1, mu2, E11, E22, E12 and ssim = A B / (C D),
    d/dmu1 = 2 mu2 (B - A)/(C D) - ssim 2 mu1 (D - C)/(C D),   d/dE11 = -ssim / D,   d/dE12 = 2 A / (C D),
    d/dx   = [ G*dmu1 + 2 x (G*dE11) + y (G*dE12) ] / n   (G symmetric, zero padded)   and   d l1/dx = sign(x - y)/n.
Pinned by tests/golden/loss_l1_ssim.npz: values and autograd gradients of the reference's own functions on CPU
(tests/golden/make_golden.py:fixture_loss).  float64 arithmetic: a checker, not a bit-level model of conv2d.
"""
import math

import numpy as np


def window1d():
    g = np.array([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=np.float32)
    return (g / g.sum(dtype=np.float32)).astype(np.float32)


def _conv(img, w2):
    """Zero-padded 11x11 correlation per channel; img (C, H, W) float64."""
    C, H, W = img.shape
    p = np.zeros((C, H + 10, W + 10))
    p[:, 5:5 + H, 5:5 + W] = img
    out = np.zeros_like(img)
    for dy in range(11):
        for dx in range(11):
            out += w2[dy, dx] * p[:, dy:dy + H, dx:dx + W]
    return out


def _moments(x, y):
    g = window1d()
    w2 = np.outer(g, g).astype(np.float32).astype(np.float64)  # _1D_window.mm(_1D_window.t()).float()
    return w2, _conv(x, w2), _conv(y, w2), _conv(x * x, w2), _conv(y * y, w2), _conv(x * y, w2)


def l1(x, y):
    return float(np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64)).mean())


def ssim(x, y):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    _, m1, m2, e11, e22, e12 = _moments(x, y)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    s1, s2, s12 = e11 - m1 * m1, e22 - m2 * m2, e12 - m1 * m2
    return float((((2 * m1 * m2 + C1) * (2 * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s1 + s2 + C2))).mean())


def grad(x, y, g_l1, g_ssim):
    """d(g_l1 * l1 + g_ssim * ssim)/dx."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    w2, m1, m2, e11, e22, e12 = _moments(x, y)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    s1, s2, s12 = e11 - m1 * m1, e22 - m2 * m2, e12 - m1 * m2
    A, B, C, D = 2 * m1 * m2 + C1, 2 * s12 + C2, m1 * m1 + m2 * m2 + C1, s1 + s2 + C2
    inv = 1.0 / (C * D)
    s = A * B * inv
    dmu = 2 * m2 * (B - A) * inv - s * 2 * m1 * (D - C) * inv
    de11 = -s / D
    de12 = 2 * A * inv
    n = x.size
    return g_l1 * np.sign(x - y) / n + g_ssim / n * (_conv(dmu, w2) + 2 * x * _conv(de11, w2) + y * _conv(de12, w2))
