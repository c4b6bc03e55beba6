"""Seeded random sweep of the rasterizer against the CPU oracle with the mode switches crossed: sizes from one Gaussian to
tens of thousands, ragged images from a single tile to hundreds, splat scales from sub-pixel to tile-filling, every SH degree,
cameras near and far, and — per case — tight / canonical lists, the colour job on / off, the direct / two-level tile sort, the
ordered backward and either launch form of preprocess_bwd.  What the fixed cases of test_gpu_raster.py pin one at a time, here in combinations nobody chose."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import raster_ref as RR  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd import rasterizer as RZ  # noqa: E402
from riggs_amd.rasterizer import rasterize_backward, saved_views  # noqa: E402
from tests import gpu_util as U  # noqa: E402

pytestmark = pytest.mark.gpu


def _case(seed):
    r = np.random.RandomState(1000 + seed)
    N = int(r.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 3001, 8000, 20000], p=[.04, .04, .04, .04, .04, .05, .05, .1, .15, .15, .15, .15]))
    H, W = int(r.randint(9, 420)), int(r.randint(9, 420))
    scale = float(np.exp(r.uniform(np.log(0.003), np.log(0.3))))
    return dict(N=N, J=int(r.choice([2, 8, 24])), H=H, W=W, scale=scale, deg=int(r.randint(0, 4)),
                radius=float(r.uniform(1.5, 5.0)), azimuth=float(r.uniform(0, 360)),
                tight=bool(r.randint(2)), jobs=bool(r.randint(2)), grouped=int(r.choice([-1, 0, 1])), ordered=bool(r.rand() < 0.25),
                bg=[float(x) for x in r.rand(3)], opacity_scale=float(r.choice([1.0, 1.0, 0.3, 0.05])), lean=int(r.choice([0, 1])))


@pytest.mark.parametrize("seed", range(40))
def test_random_configuration_matches_the_oracle(seed):
    c = _case(seed)
    sc, act, cam = U.activated_scene(c["N"], c["J"], 500 + seed, c["H"], c["W"], scale=c["scale"], radius=c["radius"],
                                     azimuth_deg=c["azimuth"])
    act["opacities"] = act["opacities"] * c["opacity_scale"]
    M = (c["deg"] + 1) ** 2
    act["shs"] = act["shs"][:, :M].contiguous()
    out_o, so = U.oracle_forward(act, cam, c["bg"], sh_degree=c["deg"])
    try:
        RZ.set_ordered_backward(c["ordered"])
        L.set_option("color_side_jobs", int(c["jobs"]))
        L.set_option("bin_grouped", int(c["grouped"]))
        L.set_option("preprocess_bwd_lean", c["lean"])
        color, radii, depth, alpha, s = U.hip_forward(act, cam, c["bg"], sh_degree=c["deg"], tight_lists=c["tight"])
        v = saved_views(s)
        assert np.array_equal(radii.cpu().numpy(), so.radii), c
        if not c["tight"]:
            U.compare_forward_state(so, v, out_o, color, depth, alpha, radii)  # (lists, ranges, depth bits: bit for bit)
        else:
            assert v["R"] <= so.R
            U.assert_close(color.cpu().numpy(), out_o["color"], "image %s" % c, U.REL_TOL, 1e-4)
            U.assert_close(alpha.cpu().numpy()[0], out_o["alpha"], "alpha %s" % c, U.REL_TOL, 1e-4)
        if so.R == 0:
            return
        g = torch.Generator().manual_seed(seed)
        gc = torch.randn(3, c["H"], c["W"], generator=g) / (3 * c["H"] * c["W"])
        gd = torch.randn(1, c["H"], c["W"], generator=g) / (c["H"] * c["W"])
        ga = torch.randn(1, c["H"], c["W"], generator=g) / (c["H"] * c["W"])
        go = RR.backward(so, gc.numpy(), gd.numpy()[0], ga.numpy()[0])
        d = lambda t: t.cuda().contiguous()  # noqa: E731
        gh = rasterize_backward(s, d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]),
                                d(act["rotations"]), None, None, None, d(gc), d(gd), d(ga))
        names = ("means3D", "means2D", "shs", None, "opacities", "scales", "rotations")
        for got, name in zip(gh, names):
            if name is None:
                continue
            want = go[name]
            # bars of the fixed cases, plus the rows of TWO Gaussians: under a pure-noise cotangent the gradient of a screen-filling
            # splat is a sum of 10^5 terms that cancels to 1e-4 of its own size, and ONE pixel whose alpha sits on the 1/255
            # threshold (a last-bit difference between expf here and on the device) moves it by per cents of itself — seed 20:
            # Gaussian 477, radius 406 px, 10 % on a value of 2e-7; with a constant cotangent the same row agrees to 3e-6
            # (tools/scratch/fuzz_case2.py)
            row = int(np.prod(want.shape[1:])) if want.ndim > 1 else 1
            U.assert_close(got.cpu().numpy().reshape(want.shape), want, "dL/d%s %s" % (name, c), 1e-4, max(1e-4, 2.0 * row / want.size))
    finally:
        RZ.set_ordered_backward(False)
        L.set_option("color_side_jobs", 1)
        L.set_option("bin_grouped", -1)
        L.set_option("preprocess_bwd_lean", -1)


# ------------------------------------------------------------------------------------------- the deformation
def _deform_case(seed):
    r = np.random.RandomState(7000 + seed)
    J = int(r.choice([2, 3, 8, 15, 16, 24, 33, 63, 64]))
    K = int(r.choice([-1, -1, 1, 2, 3, 4])) if J > 4 else -1
    return dict(N=int(r.choice([1, 63, 64, 65, 255, 257, 1023, 1025, 5000, 30011])), J=J, K=min(K, J - 1), chain=bool(r.randint(2)),
                mask=r.choice(["ones", "rand", "zeros-mixed"]), nonunit=bool(r.randint(2)))


@pytest.mark.parametrize("seed", range(24))
def test_random_deformation_matches_the_oracle(seed):
    """deform_by_pose (forward kinematics, bone-distance weights — all bones or top-K —, LBS of means and quaternions) forward
    and backward against oracle/deform_ref.py (skeleton_warp.py:41-76,130-172) on random skeletons of 2..64 joints, chains and
    trees, ragged Gaussian counts, motion masks with exact zeros, un-normalised local rotations."""
    from oracle import deform_ref as O
    from riggs_amd import synth
    from riggs_amd.skeleton import SkeletonWarp
    c = _deform_case(seed)
    N, J, K = c["N"], c["J"], c["K"]
    sc = synth.make_scene(N, J, 900 + seed, chain=c["chain"])
    g = torch.Generator().manual_seed(seed)
    gx, gr, gn = torch.randn(N, 3, generator=g), torch.randn(N, 4, generator=g), torch.randn(J, 3, generator=g)
    mask = {"ones": torch.ones(N, 1), "rand": torch.rand(N, 1, generator=g),
            "zeros-mixed": (torch.rand(N, 1, generator=g) > 0.4).float() * torch.rand(N, 1, generator=g)}[c["mask"]]
    lr = sc["local_rotation"] * ((0.5 + torch.rand(J, 1, generator=g)) if c["nonunit"] else 1.0)
    q = lr.clone().requires_grad_(True)
    gt = sc["global_trans"].clone().requires_grad_(True)
    rho = sc["node_radius"].clone().requires_grad_(True)
    mo = mask.clone().requires_grad_(True)
    o = O.deform_by_pose(sc["xyz"], sc["joints"], sc["parents"], rho, q, gt, mo, K)
    ((o["d_xyz"] * gx).sum() + (o["d_rotation"] * gr).sum() + (o["d_nodes"] * gn).sum()).backward()
    sw = SkeletonWarp(is_blender=True, joints=sc["joints"], parent_indices=sc["parents"], K=K, hyper_dim=8,
                      use_skinning_weight_mlp=False, use_template_offsets=False).cuda()
    sw._node_radius.data = sc["node_radius"].cuda()
    qh, gth = lr.cuda().requires_grad_(True), sc["global_trans"].cuda().requires_grad_(True)
    mh = mask.cuda().requires_grad_(True)
    h = sw.deform_by_pose(sc["xyz"].cuda(), {"local_rotation": qh, "global_trans": gth}, mh)
    ((h["d_xyz"] * gx.cuda()).sum() + (h["d_rotation"] * gr.cuda()).sum() + (h["d_nodes"] * gn.cuda()).sum()).backward()
    tag = str(c)
    U.assert_close(h["d_nodes"].detach().cpu().numpy(), o["d_nodes"].detach().numpy(), "d_nodes " + tag, 1e-5)
    hx, ox = h["d_xyz"].detach().cpu().numpy(), o["d_xyz"].detach().numpy()
    hr, orr = h["d_rotation"].detach().cpu().numpy(), o["d_rotation"].detach().numpy()
    if K > 0:
        # top-K: bones that share a joint are at EQUAL distance from every Gaussian whose nearest point on them is that joint, and
        # torch.topk does not define which of equal values it returns (the reference itself depends on its torch build there):
        # rows whose K-th and (K+1)-th distances are apart must select the same bones and agree in value; the others are free
        d2_all = O.bone_dist2(sc["xyz"], sc["joints"], sc["parents"]).numpy()
        srt = np.sort(d2_all, 1)
        clear = np.ones(N, bool) if K >= J - 1 else (srt[:, K] - srt[:, K - 1]) > 1e-5 * np.maximum(srt[:, K], 1e-12)
        idx, oidx = h["nn_idx"].cpu().numpy(), o["nn_idx"].numpy()
        assert np.array_equal(np.sort(idx[clear], 1), np.sort(oidx[clear], 1)), "top-K selection differs on rows without ties " + tag
        if clear.any():
            U.assert_close(hx[clear], ox[clear], "d_xyz (untied rows) " + tag)
            U.assert_close(hr[clear], orr[clear], "d_rotation (untied rows) " + tag)
        if not clear.all():
            return  # (gradients are only comparable when every row made the same choice)
    else:
        U.assert_close(hx, ox, "d_xyz " + tag)
        U.assert_close(hr, orr, "d_rotation " + tag)
    U.assert_close(qh.grad.cpu().numpy(), q.grad.numpy(), "dL/dlocal_rotation " + tag, 2e-4)
    U.assert_close(gth.grad.cpu().numpy(), gt.grad.numpy(), "dL/dglobal_trans " + tag, 2e-4)
    if J > 2:  # (one bone: its weight is 1 whatever its radius — the gradient is 0 and both sides return their own rounding noise)
        U.assert_close(sw._node_radius.grad.cpu().numpy(), rho.grad.numpy(), "dL/d_node_radius " + tag, 2e-4)
    else:
        scale_ref = float(np.abs(q.grad.numpy()).max())
        assert float(sw._node_radius.grad.abs().max()) <= 1e-4 * scale_ref and float(rho.grad.abs().max()) <= 1e-4 * scale_ref
    U.assert_close(mh.grad.cpu().numpy(), mo.grad.numpy(), "dL/dmotion_mask " + tag, 2e-4)


# ------------------------------------------------------------------------------------------- the captured frame
@pytest.mark.parametrize("seed", range(8))
def test_random_captured_frame_equals_the_eager_frame(seed):
    """GraphedFrame (hipGraph of deform -> render -> backward) at random sizes, with sparse gradient rows / tight lists / the
    colour job drawn per case, replayed over changing cameras and image gradients, against the same frames issued eagerly on a
    copy of the models: images, radii, every parameter gradient; no arena overflow, and no arena allocation inside the capture."""
    import copy

    import bench
    from riggs_amd import synth
    from riggs_amd.graph import GraphedFrame
    from riggs_amd.rasterizer import RasterArena
    r = np.random.RandomState(300 + seed)
    N, J = int(r.choice([300, 2049, 7001, 20000])), int(r.choice([4, 8, 24]))
    H, W = int(r.randint(40, 260)), int(r.randint(40, 260))
    sparse, tight, jobs = bool(r.randint(2)), bool(r.randint(2)), bool(r.randint(2))
    lean = int(r.choice([-1, 0, 1]))
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=N, J=J, H=H, W=W)
    try:
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)
    gm2, sw2 = copy.deepcopy(gm), copy.deepcopy(sw)
    bg = torch.zeros(3, device="cuda")
    try:
        L.set_option("color_side_jobs", int(jobs))
        L.set_option("preprocess_bwd_lean", lean)
        gf = GraphedFrame(gm, sw, cam, bg, bench.params_of(gm, sw), sparse_grad_rows=sparse, tight_lists=tight).capture()
        where = gf.arena.binning.data_ptr()
        arena2 = RasterArena(tight_lists=tight)
        g = torch.Generator().manual_seed(seed)
        for k in range(4):
            c = synth.look_at_camera(H, W, azimuth_deg=float(r.uniform(0, 360)), elevation_deg=float(r.uniform(-40, 60)),
                                     radius=float(r.uniform(2.5, 4.5)), fid=float(r.uniform(0, 1))).to("cuda:0")
            gimg = (torch.rand(3, H, W, generator=g) - 0.5).cuda() / (H * W)
            a = gf.run(cam=c, gimg=gimg)
            gf.check()
            step = bench.make_step(c, gm2, sw2, gimg, arena2, 1, None)
            for p in bench.params_of(gm2, sw2):
                p.grad = None
            b = step()
            torch.cuda.synchronize()
            assert torch.equal(a["radii"], b["radii"]), (seed, k)
            torch.testing.assert_close(a["render"], b["render"].detach(), rtol=1e-4, atol=2e-6)
            for ga, p2 in zip(gf.grads, bench.params_of(gm2, sw2)):
                gb = p2.grad
                torch.testing.assert_close(ga, gb, rtol=2e-4, atol=3e-6 * float(gb.abs().max()) + 1e-12)
        assert gf.arena.binning.data_ptr() == where
    finally:
        L.set_option("color_side_jobs", 1)
        L.set_option("preprocess_bwd_lean", -1)
