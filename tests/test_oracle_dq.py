"""CPU: oracle/dq_ref.py (numpy restatement of utils/dual_quaternion.py with a hand-derived backward) against golden vectors
produced by the reference's OWN functions and autograd (tests/golden/make_golden.py: fixture_dqb)."""
import glob
import os

import numpy as np
import pytest

from oracle import dq_ref as D

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "dqb_*.npz")))


def close(a, b, what, tol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    assert float(np.abs(a - b).max()) <= tol * scale, (what, float(np.abs(a - b).max()), scale)


def test_fixtures_present():
    assert len(FILES) == 7


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_oracle_matches_reference_golden(path):
    g = np.load(path)
    mode, rot_as_q = str(g["mode"]), bool(g["rot_as_q"])
    f64 = lambda k: g[k].astype(np.float64)  # noqa: E731
    if mode == "tblend":
        out = D.transformation_blending(f64("transformations"), f64("weights"))
        close(out, g["out"], "transformation_blending")
        # its gradient w.r.t. the weights, through the oracle's pieces: q(R) blend -> R(q)
        return
    if mode == "interp":
        rot, t_ = D.interpolate(f64("q0"), f64("t0"), f64("q1"), f64("t1"), f64("weight"), rot_as_q)
        close(rot, g["out_rot"], "interpolate rot")
        close(t_, g["out_t"], "interpolate t")
        # gradients: the two-node blend with weights (w, 1 - w)
        w = f64("weight")
        r, t2, cache = D.dq_blending(np.stack([f64("q0"), f64("q1")], 1), np.stack([f64("t0"), f64("t1")], 1),
                                     np.concatenate([w, 1 - w], 1), rot_as_q, norm_over_nodes=False)
        gq, gt, gw = D.dq_blending_backward(cache, f64("g_rot"), f64("g_t"))
        close(gq[:, 0], g["grad_q0"], "dL/dq0", 1e-4)
        close(gq[:, 1], g["grad_q1"], "dL/dq1", 1e-4)
        close(gt[:, 0], g["grad_t0"], "dL/dt0", 1e-4)
        close(gt[:, 1], g["grad_t1"], "dL/dt1", 1e-4)
        close(gw[:, :1] - gw[:, 1:], g["grad_weight"], "dL/dweight", 1e-4)
        return
    rot, t_, cache = D.dq_blending(f64("q"), f64("t"), f64("weights"), rot_as_q)
    close(rot.reshape(g["out_rot"].shape), g["out_rot"], "rotation")
    close(t_, g["out_t"], "translation")
    gq, gt, gw = D.dq_blending_backward(cache, f64("g_rot").reshape(rot.shape), f64("g_t"))
    close(gq, g["grad_q"], "dL/dq", 1e-4)
    close(gt, g["grad_t"], "dL/dt", 1e-4)
    close(gw, g["grad_weights"], "dL/dweights", 1e-4)


def test_the_two_reference_quirks_are_in_the_fixtures():
    """(1) a 3-D q is normalised over the NODE axis: the golden of "shared3d" differs from a per-quaternion normalisation of the
    same inputs; (2) some nodes' dual parts are sign-flipped by quaternion_multiply's standardisation."""
    g = np.load(os.path.join(GOLD, "dqb_shared3d_k63_R.npz"))
    q, t, w = (g[k].astype(np.float64) for k in ("q", "t", "weights"))
    per_quat = D.dq_blending(q, t, w, False, norm_over_nodes=False)[0]
    over_nodes = D.dq_blending(q, t, w, False, norm_over_nodes=True)[0]
    assert np.abs(over_nodes.reshape(g["out_rot"].shape) - g["out_rot"]).max() < 2e-5
    assert np.abs(per_quat - over_nodes).max() > 1e-2
    dq, cache = D.qt2dq(q, t, True)
    sgn = cache[4]
    assert (sgn < 0).any() and (sgn > 0).any()


def test_backward_against_finite_differences():
    rng = np.random.default_rng(5)
    for rot_as_q, shape in ((True, (1, 5)), (False, (6, 3))):
        N, K = 6, shape[1]
        q, t = rng.normal(size=shape + (4,)), 0.5 * rng.normal(size=shape + (3,))
        w = rng.random((N, K)) + 0.1
        g_t = rng.normal(size=(N, 3))
        rot, t_, cache = D.dq_blending(q, t, w, rot_as_q)
        g_rot = rng.normal(size=rot.shape)
        gq, gt, gw = D.dq_blending_backward(cache, g_rot, g_t)

        def loss(q_, t2, w_):
            r, tt, _ = D.dq_blending(q_, t2, w_, rot_as_q)
            return float((r * g_rot).sum() + (tt * g_t).sum())
        for arr, grad in ((q, gq), (t, gt), (w, gw)):
            for _ in range(6):
                idx = tuple(rng.integers(0, s) for s in arr.shape)
                h = 1e-6
                a1, a2 = arr.copy(), arr.copy()
                a1[idx] += h
                a2[idx] -= h
                args1 = [a1 if x is arr else x for x in (q, t, w)]
                args2 = [a2 if x is arr else x for x in (q, t, w)]
                fd = (loss(*args1) - loss(*args2)) / (2 * h)
                assert abs(fd - grad[idx]) <= 1e-5 * max(1.0, abs(fd)), (rot_as_q, idx, fd, grad[idx])
