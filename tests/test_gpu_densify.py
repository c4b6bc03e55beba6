"""-m gpu: device-side densification / pruning (riggs_amd.GaussianModel.densify_and_prune / densify_and_clone /
densify_and_split / prune_points / reset_opacity over csrc/densify.hip) against goldens of the reference's own methods and, at
training sizes, against the pinned numpy oracle; then a captured training iteration that survives a densification."""
import glob
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import densify_ref as D  # noqa: E402
from riggs_amd.gaussian_model import GaussianModel  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "densify_*.npz")))
ARGS = SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                       position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.001, rotation_lr=0.001)


def build(P, M, V, steps, accum, denom, radii2D, iso, fea_dim):
    """A riggs_amd GaussianModel on the device in a given optimizer state."""
    gm = GaussianModel(3, fea_dim=fea_dim, with_motion_mask=False, use_isotropic_gs=iso)
    T = lambda a: torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(a)).float().cuda())  # noqa: E731
    gm._xyz, gm._features_dc, gm._features_rest = T(P["xyz"]), T(P["f_dc"]), T(P["f_rest"])
    gm._opacity, gm._scaling, gm._rotation = T(P["opacity"]), T(P["scaling"]), T(P["rotation"])
    if fea_dim:
        gm.feature = T(P["feature"])
    gm.training_setup(ARGS)
    for grp in gm.optimizer.param_groups:
        p = grp["params"][0]
        k = grp["name"]
        gm.optimizer.state[p] = {"step": torch.tensor(float(steps)), "exp_avg": torch.from_numpy(M[k].copy()).cuda(),
                                 "exp_avg_sq": torch.from_numpy(V[k].copy()).cuda()}
    gm.xyz_gradient_accum, gm.denom = torch.from_numpy(accum.copy()).cuda(), torch.from_numpy(denom.copy()).cuda()
    gm.max_radii2D = torch.from_numpy(radii2D.copy()).cuda()
    return gm


def load(g, tag):
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"] + (["feature"] if int(g["fea_dim"]) else [])
    return ({k: g["%s_%s" % (tag, k)].copy() for k in names}, {k: g["%s_m_%s" % (tag, k)].copy() for k in names},
            {k: g["%s_v_%s" % (tag, k)].copy() for k in names})


def check(gm, P, M, V, stats, what, tol=2e-6):
    grp = {g_["name"]: g_ for g_ in gm.optimizer.param_groups}
    attr = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
            "rotation": "_rotation", "feature": "feature"}
    for k in P:
        p = grp[k]["params"][0]
        assert p is getattr(gm, attr[k]) and isinstance(p, torch.nn.Parameter) and p.requires_grad, (what, k)
        st = gm.optimizer.state[p]
        for got, ref, nm in ((p, P[k], "param"), (st["exp_avg"], M[k], "exp_avg"), (st["exp_avg_sq"], V[k], "exp_avg_sq")):
            assert tuple(got.shape) == ref.shape, (what, k, nm, tuple(got.shape), ref.shape)
            np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=tol, atol=1e-6 * max(1.0, float(np.abs(ref).max()) if ref.size else 1.0),
                                       err_msg="%s %s %s" % (what, k, nm))
    assert len(gm.optimizer.state) == len(P)   # no stale state entries of the replaced tensors
    if stats is not None:
        n = P["xyz"].shape[0]
        assert gm.xyz_gradient_accum.shape == (n, 1) and gm.denom.shape == (n, 1) and gm.max_radii2D.shape == (n,)
        assert np.array_equal(gm.xyz_gradient_accum.cpu().numpy(), stats["accum"]) and np.array_equal(gm.denom.cpu().numpy(), stats["denom"])
        assert np.array_equal(gm.max_radii2D.cpu().numpy(), stats["radii2D"])


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_hip_methods_match_reference_golden(path):
    g = np.load(path)
    iso, fea = bool(g["isotropic"]), int(g["fea_dim"])
    st = lambda tag: {"accum": g[tag + "_accum"], "denom": g[tag + "_denom"], "radii2D": g[tag + "_radii2D"]}  # noqa: E731
    mk = lambda: build(*load(g, "dp0"), 1.0, g["dp0_accum"], g["dp0_denom"], g["dp0_radii2D"], iso, fea)  # noqa: E731
    for tag in ("dp", "dq"):
        gm = mk()
        a = g[tag + "_args"]
        zk, _, _ = kept_children_draws(load(g, "dp0")[0], g["dp0_accum"], g["dp0_denom"], iso, g[tag + "_z"], float(a[0]), float(a[1]),
                                       float(a[2]), None if a[3] < 0 else float(a[3]))
        gm.densify_and_prune(float(a[0]), float(a[1]), float(a[2]), None if a[3] < 0 else float(a[3]), unit_normals=torch.from_numpy(zk))
        check(gm, *load(g, tag + "1"), st(tag + "1"), tag)
        for grp in gm.optimizer.param_groups:
            assert float(gm.optimizer.state[grp["params"][0]]["step"]) == 1.0
    gm = mk()
    gm.prune_points(torch.from_numpy(g["pr_mask"]).cuda())
    check(gm, *load(g, "pr1"), st("pr1"), "prune_points")
    grads = gm_grads(mk_=mk)
    gm = mk()
    gm.densify_and_clone(grads, 0.0002, 2.0)
    check(gm, *load(g, "cl1"), st("cl1"), "densify_and_clone")
    gm = mk()
    gm.densify_and_split(grads, 0.0002, 2.0, unit_normals=torch.from_numpy(g["sp_z"]))
    check(gm, *load(g, "sp1"), st("sp1"), "densify_and_split")
    gm = mk()
    gm.reset_opacity()
    check(gm, *load(g, "ro1"), None, "reset_opacity")
    # the optimizer still steps after the surgery (state keyed by the new parameters)
    for grp in gm.optimizer.param_groups:
        grp["params"][0].grad = torch.ones_like(grp["params"][0])
    gm.optimizer.step()
    assert float(gm.optimizer.state[gm._xyz]["step"]) == 2.0


def kept_children_draws(P, accum, denom, iso, z, max_grad, min_opacity, extent, screen, percent_dense=0.01):
    """The reference draws normals for the children of EVERY split parent and prunes afterwards; the device path draws only for
    the children that survive.  Picks those rows out of the reference's draws (layout: copy-major over the split parents)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        gr = np.nan_to_num(accum / denom).reshape(-1)
    smax = D.get_scaling(P, iso).max(1)
    sel = (gr >= max_grad) & (smax > percent_dense * extent)
    op = 1 / (1 + np.exp(-P["opacity"].reshape(-1).astype(np.float64)))
    kept = sel & ~((op < min_opacity) | ((smax / 1.6 > 0.1 * extent) if screen else False))
    S = int(sel.sum())
    pick = (np.cumsum(sel) - 1)[kept]
    return np.concatenate([z[:S][pick], z[S:2 * S][pick]], 0), S, int(kept.sum())


def gm_grads(mk_):
    gm = mk_()
    grads = gm.xyz_gradient_accum / gm.denom
    grads[grads.isnan()] = 0.0
    return grads


@pytest.mark.parametrize("N,iso", [(300_001, False), (70_000, True)])
def test_densify_and_prune_against_the_oracle_at_training_size(N, iso):
    rng = np.random.default_rng(N)
    f32 = np.float32
    P = {"xyz": rng.normal(size=(N, 3)).astype(f32), "f_dc": rng.normal(size=(N, 1, 3)).astype(f32),
         "f_rest": (0.1 * rng.normal(size=(N, 15, 3))).astype(f32), "opacity": (2.5 * rng.normal(size=(N, 1))).astype(f32),
         "scaling": (math.log(0.04 if iso else 0.015) + 0.9 * rng.normal(size=(N, 1 if iso else 3))).astype(f32),
         "rotation": rng.normal(size=(N, 4)).astype(f32)}
    M = {k: (0.01 * rng.normal(size=v.shape)).astype(f32) for k, v in P.items()}
    V = {k: (1e-4 * rng.random(size=v.shape)).astype(f32) for k, v in P.items()}
    denom = rng.integers(0, 4, size=(N, 1)).astype(f32)
    accum = ((rng.random((N, 1)) ** 3) * 6e-4 * denom).astype(f32)
    # keep the elements away from the thresholds (device expf / division round differently from numpy's by an ulp)
    with np.errstate(invalid="ignore", divide="ignore"):
        gr = np.nan_to_num(accum / denom)
    accum[np.abs(gr - 0.0002) < 1e-8] = 0
    with np.errstate(invalid="ignore", divide="ignore"):
        gr = np.nan_to_num(accum / denom)
    gm = build({k: v.copy() for k, v in P.items()}, M, V, 3.0, accum, denom, np.zeros(N, f32), iso, 0)
    n_children_max = 2 * N
    z = rng.normal(size=(n_children_max, 3)).astype(f32)
    Po, Mo, Vo = {k: v.copy() for k, v in P.items()}, {k: v.copy() for k, v in M.items()}, {k: v.copy() for k, v in V.items()}
    # the oracle needs exactly the draws of the children that exist: run it once to learn how many split
    smax = D.get_scaling(Po, iso).max(1)
    sel = (gr.reshape(-1) >= 0.0002) & (smax > 0.01 * 2.0)
    S = int(sel.sum())
    stats = D.densify_and_prune(Po, Mo, Vo, accum.copy(), denom.copy(), 0.0002, 0.005, 2.0, 20, 0.01, iso, z[:2 * S])
    # HIP: the children kept after the final prune are a subset of the split ones — hand it the draws of exactly those, in order
    zk, S2, n_kept = kept_children_draws(P, accum, denom, iso, z[:2 * S], 0.0002, 0.005, 2.0, 20)
    assert S2 == S
    gm.densify_and_prune(0.0002, 0.005, 2.0, 20, unit_normals=torch.from_numpy(zk))
    assert gm._xyz.shape[0] == Po["xyz"].shape[0] and gm._xyz.shape[0] != N
    check(gm, Po, Mo, Vo, stats, "densify_and_prune at %d" % N, tol=1e-5)
    assert S > 100 and n_kept < S  # some parents' children do not survive the final prune: the draws were re-indexed


def test_captured_training_iteration_survives_a_device_side_densification():
    """GraphedTrainStep -> GaussianModel.densify_and_prune (new parameter tensors, new N) -> recapture(): the iteration goes
    on with the densified cloud and the carried-over moments (train_rig.py:359-365 followed by the next iterations)."""
    from riggs_amd import synth
    from riggs_amd.graph import GraphedTrainStep
    from riggs_amd.optim import FusedAdam
    from riggs_amd.skeleton import SkeletonWarp
    N, J, H, W = 20_000, 8, 128, 128
    sc = synth.make_scene(N, J, 7, scale=0.03)
    cam = synth.look_at_camera(H, W, fid=0.3).to("cuda")
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"], sc["opacity"])
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8, use_skinning_weight_mlp=False,
                      use_template_offsets=False).cuda()
    sw._node_radius.data = sc["node_radius"].cuda()
    gm.training_setup(ARGS, capturable=True)
    sk = FusedAdam([{"params": g_["params"], "lr": 5e-4, "name": g_["name"]} for g_ in sw.trainable_parameters()], lr=0.0, eps=1e-15, capturable=True)
    target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device="cuda"), target, [gm.optimizer, sk], lambda_dssim=0.2).capture()
    for _ in range(5):
        out = gts.run()
        gm.add_densification_stats(SimpleNamespace(grad=out["viewspace_points_grad"]), out["radii"] > 0, out["radii"])
    torch.cuda.synchronize()
    l0 = float(out["loss"])
    m_before = gm.optimizer.state[gm._xyz]["exp_avg"].clone()
    n0 = gm._xyz.shape[0]
    gm.densify_and_prune(1e-7, 0.005, 2.0, 20)
    n1 = gm._xyz.shape[0]
    assert n1 != n0 and gm.optimizer.state[gm._xyz]["exp_avg"].shape[0] == n1
    assert float(gm.optimizer.state[gm._xyz]["exp_avg"].abs().sum()) > 0 and float(m_before.abs().sum()) > 0
    gts.recapture()
    for _ in range(5):
        out = gts.run()
    torch.cuda.synchronize()
    assert out["radii"].shape[0] == n1 and math.isfinite(float(out["loss"])) and float(out["loss"]) < 1.5 * l0
    assert float(gm.optimizer.state[gm._xyz]["step"]) == 2.0 + 5.0 + 1.0 + 5.0  # the eager warm-up frames of a capture ARE iterations
