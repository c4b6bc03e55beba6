"""ORACLE (test infrastructure, NOT product code).

Densification / pruning of the Gaussian cloud: a numpy restatement of /root/reference/scene/gaussian_model.py —
densify_and_prune :500-514, densify_and_clone :475-498, densify_and_split :440-473, prune_points :373-392,
densification_postfix :419-438, the optimizer surgery _prune_optimizer :355-371 / cat_tensors_to_optimizer :394-417 and
reset_opacity :275-278 — on plain dicts of arrays: ``P`` (parameters by group name), ``M`` / ``V`` (Adam moments).
Only ``tests/`` may import this module.  Parity status: PINNED by tests/golden/densify_*.npz = the reference's own methods
run on the CPU (tests/golden/make_golden.py: fixture_densify), checked by tests/test_oracle_densify.py.
"""
from __future__ import annotations

import numpy as np


def get_scaling(P, isotropic):
    s = np.exp(P["scaling"])
    return np.repeat(s[:, :1], 3, 1) if isotropic else s          # :104-110


def build_rotation(r):
    """utils/general_utils.py:137-158."""
    q = r / np.sqrt((r * r).sum(1, keepdims=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def _cat(P, M, V, new):
    """cat_tensors_to_optimizer (:394-417) + densification_postfix (:419-438): new rows appended, their moments zero."""
    for k in P:
        P[k] = np.concatenate([P[k], new[k]], 0)
        M[k] = np.concatenate([M[k], np.zeros_like(new[k])], 0)
        V[k] = np.concatenate([V[k], np.zeros_like(new[k])], 0)
    n = P["xyz"].shape[0]
    return {"accum": np.zeros((n, 1), np.float32), "denom": np.zeros((n, 1), np.float32), "radii2D": np.zeros(n, np.float32)}


def prune_points(P, M, V, stats, mask):
    keep = ~mask                                                     # :373-392
    for k in P:
        P[k], M[k], V[k] = P[k][keep], M[k][keep], V[k][keep]
    return {k: v[keep] for k, v in stats.items()}


def densify_and_clone(P, M, V, grads, grad_threshold, dense_limit, isotropic):
    sel = (np.abs(grads.reshape(-1)) >= grad_threshold) & (get_scaling(P, isotropic).max(1) <= dense_limit)   # :478-481
    return _cat(P, M, V, {k: v[sel] for k, v in P.items()})


def densify_and_split(P, M, V, grads, grad_threshold, dense_limit, isotropic, z, N=2):
    n = P["xyz"].shape[0]
    padded = np.zeros(n, np.float32)
    padded[:grads.shape[0]] = grads.reshape(-1)                     # :444-446
    sel = (padded >= grad_threshold) & (get_scaling(P, isotropic).max(1) > dense_limit)
    stds = np.tile(get_scaling(P, isotropic)[sel], (N, 1))
    samples = stds * z                                               # torch.normal(0, stds) = stds * z
    rots = np.tile(build_rotation(P["rotation"][sel]), (N, 1, 1))
    new = {k: np.tile(v[sel], (N,) + (1,) * (v.ndim - 1)) for k, v in P.items()}
    new["xyz"] = np.einsum("nij,nj->ni", rots, samples) + np.tile(P["xyz"][sel], (N, 1))
    new["scaling"] = np.log(np.tile(get_scaling(P, isotropic)[sel], (N, 1)) / (0.8 * N))
    if isotropic:
        new["scaling"] = new["scaling"][:, :1]
    stats = _cat(P, M, V, new)
    return prune_points(P, M, V, stats, np.concatenate([sel, np.zeros(N * int(sel.sum()), bool)]))   # :469-473


def densify_and_prune(P, M, V, accum, denom, max_grad, min_opacity, extent, max_screen_size, percent_dense, isotropic, z):
    with np.errstate(invalid="ignore", divide="ignore"):
        grads = accum / denom
    grads[np.isnan(grads)] = 0.0                                     # :501-502
    densify_and_clone(P, M, V, grads, max_grad, percent_dense * extent, isotropic)
    stats = densify_and_split(P, M, V, grads, max_grad, percent_dense * extent, isotropic, z)
    prune = (1.0 / (1.0 + np.exp(-P["opacity"])) < min_opacity).reshape(-1)
    if max_screen_size:
        prune = prune | (stats["radii2D"] > max_screen_size) | (get_scaling(P, isotropic).max(1) > 0.1 * extent)   # :508-512
    return prune_points(P, M, V, stats, prune)


def reset_opacity(P, M, V):
    op = np.minimum(1.0 / (1.0 + np.exp(-P["opacity"])), 0.01)     # :275-278
    P["opacity"] = np.log(op / (1 - op)).astype(np.float32)
    M["opacity"], V["opacity"] = np.zeros_like(P["opacity"]), np.zeros_like(P["opacity"])
