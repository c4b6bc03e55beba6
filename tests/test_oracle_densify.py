"""CPU: oracle/densify_ref.py against the goldens of the reference's own densify_and_prune / prune_points / densify_and_clone /
densify_and_split / reset_opacity (tests/golden/make_golden.py: fixture_densify)."""
import glob
import os

import numpy as np
import pytest

from oracle import densify_ref as D

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "densify_*.npz")))


def load(g, tag):
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"] + (["feature"] if int(g["fea_dim"]) else [])
    P = {k: g["%s_%s" % (tag, k)].copy() for k in names}
    M = {k: g["%s_m_%s" % (tag, k)].copy() for k in names}
    V = {k: g["%s_v_%s" % (tag, k)].copy() for k in names}
    return P, M, V


def same(P, M, V, stats, g, tag):
    for k in P:
        for got, key in ((P[k], "%s_%s"), (M[k], "%s_m_%s"), (V[k], "%s_v_%s")):
            ref = g[key % (tag, k)]
            assert got.shape == ref.shape, (tag, k, got.shape, ref.shape)
            np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-7, err_msg="%s %s" % (tag, k))
    if stats is not None:
        assert np.array_equal(stats["accum"], g[tag + "_accum"]) and np.array_equal(stats["denom"], g[tag + "_denom"])
        assert np.array_equal(stats["radii2D"], g[tag + "_radii2D"])


def test_fixtures_present():
    assert len(FILES) == 2


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_oracle_matches_reference_golden(path):
    g = np.load(path)
    iso, pd = bool(g["isotropic"]), float(g["percent_dense"])
    for tag in ("dp", "dq"):
        P, M, V = load(g, tag + "0")
        a = g[tag + "_args"]
        stats = D.densify_and_prune(P, M, V, g[tag + "0_accum"].copy(), g[tag + "0_denom"].copy(), float(a[0]), float(a[1]), float(a[2]),
                                    None if a[3] < 0 else float(a[3]), pd, iso, g[tag + "_z"])
        same(P, M, V, stats, g, tag + "1")
        assert P["xyz"].shape[0] != g[tag + "0_xyz"].shape[0]
    P, M, V = load(g, "dp0")
    st0 = {"accum": g["dp0_accum"], "denom": g["dp0_denom"], "radii2D": g["dp0_radii2D"]}
    same(P, M, V, D.prune_points(P, M, V, st0, g["pr_mask"]), g, "pr1")
    with np.errstate(invalid="ignore", divide="ignore"):
        grads = g["dp0_accum"] / g["dp0_denom"]
    grads[np.isnan(grads)] = 0
    P, M, V = load(g, "dp0")
    same(P, M, V, D.densify_and_clone(P, M, V, grads, 0.0002, pd * 2.0, iso), g, "cl1")
    P, M, V = load(g, "dp0")
    same(P, M, V, D.densify_and_split(P, M, V, grads, 0.0002, pd * 2.0, iso, g["sp_z"]), g, "sp1")
    P, M, V = load(g, "dp0")
    D.reset_opacity(P, M, V)
    same(P, M, V, None, g, "ro1")
    assert float(g["ro1_step_opacity"]) == float(g["dp0_step_opacity"]) == 1.0   # the step count survives the surgery
