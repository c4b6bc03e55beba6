"""The optimizer oracle against vectors produced by the reference's own optimizer (tests/golden/make_golden.py)."""
import os

import numpy as np

from oracle import optim_ref as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "optim_adam_n67.npz"))
NAMES = [str(n) for n in G["names"]]


def test_adam_matches_reference_optimizer():
    b1, b2 = (float(x) for x in G["betas"])
    eps = float(G["eps"])
    state = {n: (G["p0_" + n], np.zeros_like(G["p0_" + n]), np.zeros_like(G["p0_" + n])) for n in NAMES}
    for it in range(int(G["steps"])):
        for gi, n in enumerate(NAMES):
            p, m, v = state[n]
            p, m, v = O.adam_step(p, G["g%d_%s" % (it, n)], m, v, it + 1, float(G["lr%d" % it][gi]), b1, b2, eps)
            state[n] = (p, m, v)
            # float32 op-for-op restatement: identical up to FMA contraction inside torch's vectorised CPU kernels
            # (visible only where m + 0.1 (g - m) cancels: absolute slack of 2e-7 of the tensor's scale)
            for got, key in ((m, "m"), (v, "v"), (p, "p")):
                want = G["%s%d_%s" % (key, it + 1, n)]
                np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-7 * float(np.abs(want).max()))
    # the xyz learning rate really was rescheduled between the steps (update_learning_rate)
    assert G["lr1"][0] < G["lr0"][0]


def test_densification_stats_match_reference():
    N = int(G["N"])
    acc, den = np.zeros((N, 1), np.float32), np.zeros((N, 1), np.float32)
    for it in range(2):
        acc, den = O.densification_stats(G["ds_grad%d" % it], G["ds_filter%d" % it], acc, den)
        np.testing.assert_allclose(acc, G["ds_accum%d" % it], rtol=1e-6, atol=0)
        np.testing.assert_array_equal(den, G["ds_denom%d" % it])


def test_training_setup_groups_and_schedule_on_cpu():
    """The host mirror of GaussianModel.training_setup / update_learning_rate (no GPU needed: nothing is stepped)."""
    from types import SimpleNamespace

    import torch

    from riggs_amd.gaussian_model import GaussianModel
    args = SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                           position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05,
                           scaling_lr=0.001, rotation_lr=0.001, skeleton_gs_position_lr=0.00001)
    t = lambda n: torch.from_numpy(G["p0_" + n])  # noqa: E731
    gm = GaussianModel.from_tensors(t("xyz"), t("f_dc"), t("f_rest"), t("scaling"), t("rotation"), t("opacity"), device="cpu")
    gm.training_setup(args)
    assert [g["name"] for g in gm.optimizer.param_groups] == NAMES
    assert isinstance(gm.optimizer, torch.optim.Adam) and gm.optimizer.defaults["eps"] == 1e-15
    for it in range(int(G["steps"])):
        np.testing.assert_allclose([g["lr"] for g in gm.optimizer.param_groups], G["lr%d" % it], rtol=1e-12)
        gm.update_learning_rate(1000 * (it + 1))
