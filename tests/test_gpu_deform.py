"""-m gpu: HIP skeleton deformation vs (a) golden vectors captured from the REAL RigGS reference
and (b) the pinned CPU oracle on larger seeded inputs.  Tolerance: 1e-4 relative (north_star)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import deform_ref as O  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402
from tests import gpu_util as U  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEFORM = sorted(glob.glob(os.path.join(GOLD, "deform_*.npz")))


def T(a, dev="cuda"):
    return torch.from_numpy(np.asarray(a)).to(dev)


def make_warp(joints, parents, rho, K):
    sw = SkeletonWarp(is_blender=True, joints=joints, parent_indices=parents, K=K, hyper_dim=8,
                      use_skinning_weight_mlp=False, use_template_offsets=False).cuda()
    sw._node_radius.data = rho.cuda()
    return sw


@pytest.mark.parametrize("path", DEFORM, ids=[os.path.basename(p)[:-4] for p in DEFORM])
def test_deform_matches_reference_golden(path):
    g = np.load(path)
    K = int(g["K"])
    sw = make_warp(T(g["joints"], "cpu"), T(g["parents"], "cpu"), T(g["node_radius_log"], "cpu"), K)
    q = T(g["local_rot"]).clone().requires_grad_(True)
    gt = T(g["global_trans"]).clone().requires_grad_(True)
    mask = T(g["motion_mask"]).clone().requires_grad_(True)
    out = sw.deform_by_pose(T(g["x"]), {"local_rotation": q, "global_trans": gt}, mask)
    if K > 0:
        # top-K: bones sharing a joint give distances that tie up to rounding; rows whose K-th / (K+1)-th
        # reference distances are separated must select the same bones, the (few) tied rows may legitimately pick
        # either bone (torch.topk itself does not define the tie order).
        d2_all = O.bone_dist2(T(g["x"], "cpu"), T(g["joints"], "cpu"), T(g["parents"], "cpu")).numpy()
        srt = np.sort(d2_all, 1)
        clear = (srt[:, K] - srt[:, K - 1]) > 1e-5 * np.maximum(srt[:, K], 1e-12)
        assert clear.mean() > 0.8
        idx = out["nn_idx"].cpu().numpy()
        same = np.array_equal(np.sort(idx[clear], 1), np.sort(g["nn_idx"][clear], 1))
        assert same, "top-K selection differs on rows without ties"
        rows = clear & np.all(np.sort(idx, 1) == np.sort(g["nn_idx"], 1), axis=1)
        U.assert_close(out["d_xyz"].detach().cpu().numpy()[rows], g["d_xyz"][rows], "d_xyz (K=%d)" % K)
        U.assert_close(out["d_rotation"].detach().cpu().numpy()[rows], g["d_rotation"][rows], "d_rotation")
        wh = np.take_along_axis(out["nn_weight"].cpu().numpy(), np.argsort(idx, 1), 1)
        wr = np.take_along_axis(g["nn_weight"], np.argsort(g["nn_idx"], 1), 1)
        U.assert_close(wh[rows], wr[rows], "nn_weight")
        if rows.all():  # gradients are only comparable when every row made the same choice
            loss = (out["d_xyz"] * T(g["g_xyz"])).sum() + (out["d_rotation"] * T(g["g_rot"])).sum() \
                + (out["d_nodes"] * T(g["g_nodes"])).sum()
            loss.backward()
            U.assert_close(q.grad.cpu().numpy(), g["grad_local_rot"], "dL/dlocal_rotation")
            U.assert_close(sw._node_radius.grad.cpu().numpy(), g["grad_node_radius"], "dL/d_node_radius")
        return
    U.assert_close(out["d_xyz"].detach().cpu().numpy(), g["d_xyz"], "d_xyz")
    U.assert_close(out["d_rotation"].detach().cpu().numpy(), g["d_rotation"], "d_rotation")
    U.assert_close(out["d_nodes"].detach().cpu().numpy(), g["d_nodes"], "d_nodes", 1e-5)
    assert float(out["d_scaling"].abs().max()) == 0.0
    assert out["d_opacity"] is None and out["d_color"] is None
    idx = out["nn_idx"].cpu().numpy()
    assert np.array_equal(idx, g["nn_idx"])
    U.assert_close(out["nn_weight"].cpu().numpy(), g["nn_weight"], "nn_weight")
    loss = (out["d_xyz"] * T(g["g_xyz"])).sum() + (out["d_rotation"] * T(g["g_rot"])).sum() \
        + (out["d_nodes"] * T(g["g_nodes"])).sum()
    loss.backward()
    U.assert_close(q.grad.cpu().numpy(), g["grad_local_rot"], "dL/dlocal_rotation")
    U.assert_close(gt.grad.cpu().numpy(), g["grad_global_trans"], "dL/dglobal_trans")
    U.assert_close(sw._node_radius.grad.cpu().numpy(), g["grad_node_radius"], "dL/d_node_radius")
    U.assert_close(mask.grad.cpu().numpy(), g["grad_motion_mask"], "dL/dmotion_mask")


@pytest.mark.parametrize("N,J,chain,seed", [(10_000, 8, True, 1235), (150_001, 24, False, 1236), (40_000, 64, False, 3)])
def test_deform_matches_oracle_large(N, J, chain, seed):
    sc = synth.make_scene(N, J, seed, chain=chain)
    g = torch.Generator().manual_seed(seed)
    gx, gr, gn = torch.randn(N, 3, generator=g), torch.randn(N, 4, generator=g), torch.randn(J, 3, generator=g)
    # oracle (torch CPU autograd)
    q = sc["local_rotation"].clone().requires_grad_(True)
    gt = sc["global_trans"].clone().requires_grad_(True)
    rho = sc["node_radius"].clone().requires_grad_(True)
    o = O.deform_by_pose(sc["xyz"], sc["joints"], sc["parents"], rho, q, gt, sc["motion_mask"], -1)
    ((o["d_xyz"] * gx).sum() + (o["d_rotation"] * gr).sum() + (o["d_nodes"] * gn).sum()).backward()
    # HIP
    sw = make_warp(sc["joints"], sc["parents"], sc["node_radius"], -1)
    qh = sc["local_rotation"].cuda().requires_grad_(True)
    gth = sc["global_trans"].cuda().requires_grad_(True)
    h = sw.deform_by_pose(sc["xyz"].cuda(), {"local_rotation": qh, "global_trans": gth}, sc["motion_mask"].cuda())
    ((h["d_xyz"] * gx.cuda()).sum() + (h["d_rotation"] * gr.cuda()).sum() + (h["d_nodes"] * gn.cuda()).sum()).backward()
    U.assert_close(h["d_xyz"].detach().cpu().numpy(), o["d_xyz"].detach().numpy(), "d_xyz")
    U.assert_close(h["d_rotation"].detach().cpu().numpy(), o["d_rotation"].detach().numpy(), "d_rotation")
    U.assert_close(h["d_nodes"].detach().cpu().numpy(), o["d_nodes"].detach().numpy(), "d_nodes", 1e-5)
    U.assert_close(qh.grad.cpu().numpy(), q.grad.numpy(), "dL/dlocal_rotation", 2e-4)
    U.assert_close(gth.grad.cpu().numpy(), gt.grad.numpy(), "dL/dglobal_trans", 2e-4)
    U.assert_close(sw._node_radius.grad.cpu().numpy(), rho.grad.numpy(), "dL/d_node_radius", 2e-4)


@pytest.mark.parametrize("N,J,two", [(1_000_001, 16, True), (1_000_000, 15, False), (999_999, 64, False), (2_000_000, 64, True)])
def test_deform_forward_values_vs_oracle_at_both_sides_of_the_two_per_lane_switch(N, J, two):
    """The skinning forward takes TWO Gaussians per lane from a million Gaussians and 16 joints on (csrc/deform.hip:
    LBS_PTS2_MIN_N / LBS_PTS2_MIN_J — the C5 path): its output VALUES against oracle/deform_ref.py (which restates
    skeleton_warp.py:130-172) at sizes on both sides of the switch, with a ragged last block, through both entry points —
    riggs_lbs_forward (deform_by_pose: FK as its own launch) and riggs_lbs_forward_fk (SkeletonWarp.forward: FK inside).  The
    oracle runs in slices of 125k Gaussians (the deformation of a Gaussian depends on the pose alone)."""
    assert (N >= 1_000_000 and J >= 16) == two
    sc = synth.make_scene(N, J, 77 + J)
    g = torch.Generator().manual_seed(N % 1000 + J)
    mask = torch.rand(N, 1, generator=g)  # a real motion mask, not ones
    sw = make_warp(sc["joints"], sc["parents"], sc["node_radius"], -1)
    x = sc["xyz"].cuda()
    with torch.no_grad():
        h = sw.deform_by_pose(x, {"local_rotation": sc["local_rotation"].cuda(), "global_trans": sc["global_trans"].cuda()}, mask.cuda())
        t = sw.expand_time(torch.tensor([0.41], device="cuda"))
        f = sw(x, t, motion_mask=mask.cuda())
    torch.cuda.synchronize()
    for tag, got, q, gt in (("two launches", h, sc["local_rotation"], sc["global_trans"]),
                            ("FK inside", f, f["local_rotation"].cpu(), f["global_trans"].cpu().reshape(-1))):
        dx, dr = got["d_xyz"].cpu().numpy(), got["d_rotation"].cpu().numpy()
        ox, orr = np.empty_like(dx), np.empty_like(dr)
        with torch.no_grad():
            for a in range(0, N, 125_000):
                b = min(N, a + 125_000)
                o = O.deform_by_pose(sc["xyz"][a:b], sc["joints"], sc["parents"], sc["node_radius"], q, gt, mask[a:b], -1)
                ox[a:b], orr[a:b] = o["d_xyz"].numpy(), o["d_rotation"].numpy()
        U.assert_close(dx, ox, "d_xyz (%s, N=%d, J=%d)" % (tag, N, J), 2e-5)
        U.assert_close(dr, orr, "d_rotation (%s)" % tag, 2e-5)
        U.assert_close(got["d_nodes"].cpu().numpy(), o["d_nodes"].numpy(), "d_nodes (%s)" % tag, 1e-5)
        # the last, ragged block and the second Gaussian of every lane pair are the places a two-per-lane indexing slip would
        # show: compare them on their own so that a failure names them
        U.assert_close(dx[-700:], ox[-700:], "d_xyz tail (%s)" % tag, 2e-5)
        U.assert_close(dx[256:512], ox[256:512], "d_xyz second half of the first 512-block (%s)" % tag, 2e-5)


@pytest.mark.parametrize("N,J", [(1, 2), (257, 8), (40_003, 24), (70_001, 64)])
def test_scalar_bone_records_give_the_same_bits_as_the_lds_form(N, J):
    """riggs_set_option("lbs_scalar", 1): the all-bones skinning forward with the bone records read through the scalar cache (a
    one-workgroup table launch in front — with the kinematic chain inside it in the _fk entry) — against the LDS form
    ("lbs_scalar" -1): identical bits in d_xyz / d_rotation / d_nodes and the chain's outputs, through both entry points; the default
    (0) picks by size and is what the million-Gaussian tests above run."""
    from riggs_amd import _lib as L
    sc = synth.make_scene(N, J, 5 + J)
    mask = torch.rand(N, 1, generator=torch.Generator().manual_seed(N)).cuda()
    sw = make_warp(sc["joints"], sc["parents"], sc["node_radius"], -1)
    x = sc["xyz"].cuda()
    res = {}
    try:
        for flag in (-1, 1):
            L.set_option("lbs_scalar", flag)
            with torch.no_grad():
                h = sw.deform_by_pose(x, {"local_rotation": sc["local_rotation"].cuda(), "global_trans": sc["global_trans"].cuda()}, mask)
                f = sw(x, sw.expand_time(torch.tensor([0.41], device="cuda")), motion_mask=mask)
            res[flag] = [t.clone() for t in (h["d_xyz"], h["d_rotation"], h["d_nodes"], f["d_xyz"], f["d_rotation"], f["d_nodes"],
                                             f["local_rotation"])]
    finally:
        L.set_option("lbs_scalar", 0)
    for a, b in zip(res[-1], res[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("keep", [0.0, 0.05, 0.5])
def test_deform_backward_with_sparse_incoming_gradient(keep):
    """The LBS backward walks only the Gaussians whose incoming gradient is non-zero (most of a deep scene gets none
    from the rasterizer): values, per-Gaussian outputs and parameter gradients against the CPU oracle for cotangents
    that are zero for most (or all) Gaussians, in odd-sized blocks."""
    N, J, seed = 70_001, 24, 21
    sc = synth.make_scene(N, J, seed)
    g = torch.Generator().manual_seed(seed)
    m = (torch.rand(N, 1, generator=g) < keep).float()
    gx, gr = torch.randn(N, 3, generator=g) * m, torch.randn(N, 4, generator=g) * m
    q = sc["local_rotation"].clone().requires_grad_(True)
    gt = sc["global_trans"].clone().requires_grad_(True)
    rho = sc["node_radius"].clone().requires_grad_(True)
    mk = sc["motion_mask"].clone().requires_grad_(True)
    o = O.deform_by_pose(sc["xyz"], sc["joints"], sc["parents"], rho, q, gt, mk, -1)
    ((o["d_xyz"] * gx).sum() + (o["d_rotation"] * gr).sum()).backward()
    sw = make_warp(sc["joints"], sc["parents"], sc["node_radius"], -1)
    qh = sc["local_rotation"].cuda().requires_grad_(True)
    gth = sc["global_trans"].cuda().requires_grad_(True)
    mkh = sc["motion_mask"].cuda().requires_grad_(True)
    h = sw.deform_by_pose(sc["xyz"].cuda(), {"local_rotation": qh, "global_trans": gth}, mkh)
    ((h["d_xyz"] * gx.cuda()).sum() + (h["d_rotation"] * gr.cuda()).sum()).backward()
    U.assert_close(qh.grad.cpu().numpy(), q.grad.numpy(), "dL/dlocal_rotation", 2e-4)
    U.assert_close(gth.grad.cpu().numpy(), gt.grad.numpy(), "dL/dglobal_trans", 2e-4)
    U.assert_close(sw._node_radius.grad.cpu().numpy(), rho.grad.numpy(), "dL/d_node_radius", 2e-4)
    U.assert_close(mkh.grad.cpu().numpy().reshape(-1), mk.grad.numpy().reshape(-1), "dL/dmotion_mask", 2e-4)
    if keep == 0.0:
        assert float(qh.grad.abs().max()) == 0.0 and float(mkh.grad.abs().max()) == 0.0


def test_forward_through_pose_net_and_node_deformation():
    sc = synth.make_scene(2048, 24, 5)
    sw = make_warp(sc["joints"], sc["parents"], sc["node_radius"], -1)
    t = sw.expand_time(torch.tensor([0.37], device="cuda"))
    assert t.shape == (24, 1)
    out = sw(sc["xyz"].cuda(), t, motion_mask=sc["motion_mask"].cuda())
    pose = sw.get_pose_info(t)
    ref = O.deform_by_pose(sc["xyz"], sc["joints"], sc["parents"], sc["node_radius"], pose["local_rotation"].detach().cpu(),
                           pose["global_trans"].detach().cpu(), sc["motion_mask"], -1)
    U.assert_close(out["d_xyz"].detach().cpu().numpy(), ref["d_xyz"].numpy(), "d_xyz via pose_net")
    out["d_xyz"].sum().backward()
    assert all(p.grad is not None for p in sw.pose_net.parameters())
    nd = sw.node_deformation(sc["joints"].cuda(), pose)
    U.assert_close(nd["d_xyz"].detach().cpu().numpy(), (ref["d_nodes"] - sc["joints"]).numpy(), "node_deformation", 1e-5)


def test_heads_with_topk_fail_loudly():
    """use_skinning_weight_mlp with K > 0 is ill-defined in the reference (1-based gather off the (N, J-1) columns)."""
    sc = synth.make_scene(64, 8, 1)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=3).cuda()  # reference defaults: both heads on
    with pytest.raises(NotImplementedError):
        sw.deform_by_pose(sc["xyz"].cuda(), {"local_rotation": sc["local_rotation"].cuda(),
                                             "global_trans": sc["global_trans"].cuda()}, sc["motion_mask"].cuda())


def test_mlp_heads_match_reference_golden():
    """both per-Gaussian heads on (the shipped stage-2 recipe) — the HIP skinning kernels take
    sigmoid(WeightMLP(x)) as weight_mod and return dL/dweight_mod; values and every gradient (pose, radii, head
    parameters) against the reference's own deform_by_pose."""
    import os
    from tests.test_oracle_heads import seeded_heads
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "heads_tree12_n200.npz"))
    T = torch.from_numpy
    J = G["joints"].shape[0]
    sw = SkeletonWarp(joints=T(G["joints"]), parent_indices=T(G["parents"]), K=-1, hyper_dim=8)
    sw.skinning_weight_mlp, sw.detail_net = seeded_heads(J, int(G["head_seed"]))
    sw = sw.cuda()
    sw._node_radius.data = T(G["node_radius"]).cuda()
    assert {g["name"] for g in sw.trainable_parameters()} == {"nodes", "pose", "skinning_mlp", "detail_net"}
    q = T(G["local_rot"]).cuda().requires_grad_(True)
    gt = T(G["global_trans"]).cuda().requires_grad_(True)
    out = sw.deform_by_pose(T(G["x"]).cuda(), {"local_rotation": q, "global_trans": gt}, T(G["mask"]).cuda())
    U.assert_close(out["d_xyz"].detach().cpu().numpy(), G["d_xyz"], "d_xyz with heads", 1e-4)
    U.assert_close(out["d_rotation"].detach().cpu().numpy(), G["d_rotation"], "d_rotation with heads", 1e-4)
    U.assert_close(out["nn_weight"].cpu().numpy(), G["nn_weight"], "nn_weight with heads", 1e-4)
    U.assert_close(sw.template_offsets.detach().cpu().numpy(), G["template_offsets"], "template offsets", 1e-4)
    ((out["d_xyz"] * T(G["c_xyz"]).cuda()).sum() + (out["d_rotation"] * T(G["c_rot"]).cuda()).sum()).backward()
    for got, key in ((q.grad, "g_local_rot"), (gt.grad, "g_global_trans"), (sw._node_radius.grad, "g_node_radius")):
        U.assert_close(got.cpu().numpy(), G[key], key, 3e-4)
    sd = {"wm": dict(sw.skinning_weight_mlp.named_parameters()), "dn": dict(sw.detail_net.named_parameters())}
    for k in G.files:
        if k.startswith("g_wm_") or k.startswith("g_dn_"):
            U.assert_close(sd[k[2:4]][k[5:]].grad.cpu().numpy(), G[k], k, 2e-3)


def test_mlp_heads_with_sparse_incoming_gradient_match_oracle():
    """Heads on (weight_mod through the HIP skinning kernels) and a cotangent that is zero for most Gaussians: the
    kernels skip those Gaussians and must still deliver exact zeros in dL/dweight_mod for them — every gradient,
    incl. the WeightMLP parameters', against the CPU oracle (which is pinned to the reference by the fixture)."""
    import os
    from tests.test_oracle_heads import seeded_heads
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "heads_tree12_n200.npz"))
    T = torch.from_numpy
    J = G["joints"].shape[0]
    g = torch.Generator().manual_seed(5)
    N = 5_003
    x = torch.randn(N, 3, generator=g) * 0.4
    keepm = (torch.rand(N, 1, generator=g) < 0.1).float()
    cx, cr = torch.randn(N, 3, generator=g) * keepm, torch.randn(N, 4, generator=g) * keepm
    mask = torch.rand(N, 1, generator=g)
    # oracle
    wm, dn = seeded_heads(J, int(G["head_seed"]))
    q = T(G["local_rot"]).clone().requires_grad_(True)
    gt = T(G["global_trans"]).clone().requires_grad_(True)
    rho = T(G["node_radius"]).clone().requires_grad_(True)
    pose = q.detach().reshape(-1)[None].expand(N, -1)
    o = O.deform_by_pose(x, T(G["joints"]), T(G["parents"]), rho, q, gt, mask, -1,
                         template_offsets=dn(x, pose), weight_offsets=wm(x))
    ((o["d_xyz"] * cx).sum() + (o["d_rotation"] * cr).sum()).backward()
    # HIP
    sw = SkeletonWarp(joints=T(G["joints"]), parent_indices=T(G["parents"]), K=-1, hyper_dim=8)
    sw.skinning_weight_mlp, sw.detail_net = seeded_heads(J, int(G["head_seed"]))
    sw = sw.cuda()
    sw._node_radius.data = T(G["node_radius"]).cuda()
    qh = T(G["local_rot"]).cuda().requires_grad_(True)
    gth = T(G["global_trans"]).cuda().requires_grad_(True)
    h = sw.deform_by_pose(x.cuda(), {"local_rotation": qh, "global_trans": gth}, mask.cuda())
    U.assert_close(h["d_xyz"].detach().cpu().numpy(), o["d_xyz"].detach().numpy(), "d_xyz with heads", 1e-4)
    ((h["d_xyz"] * cx.cuda()).sum() + (h["d_rotation"] * cr.cuda()).sum()).backward()
    U.assert_close(qh.grad.cpu().numpy(), q.grad.numpy(), "dL/dlocal_rotation", 3e-4)
    U.assert_close(gth.grad.cpu().numpy(), gt.grad.numpy(), "dL/dglobal_trans", 3e-4)
    U.assert_close(sw._node_radius.grad.cpu().numpy(), rho.grad.numpy(), "dL/d_node_radius", 3e-4)
    ref = dict(wm.named_parameters())
    for name, p in sw.skinning_weight_mlp.named_parameters():
        U.assert_close(p.grad.cpu().numpy(), ref[name].grad.numpy(), "WeightMLP " + name, 2e-3)


@pytest.mark.parametrize("width,J", [(256, 24), (32, 24), (64, 8)])
def test_fused_pose_mlp_matches_torch_and_reference_fixture(width, J):
    """riggs_pose_mlp_* (3 HIP launches) vs the torch-op PoseMLP (CPU) — outputs and every parameter gradient;
    for width 32 also vs the golden vector captured from the reference's own PoseMLP."""
    import copy
    from riggs_amd.skeleton import PoseMLP
    torch.manual_seed(5)
    net = PoseMLP(1, J * 4, depth=8, hidden_dimensions=width, multires=8)
    t = torch.tensor([0.37])
    if width == 32:
        g = np.load(os.path.join(GOLD, "posemlp_w32_j24.npz"))
        net.load_state_dict({k.replace("__", "."): torch.from_numpy(g[k]) for k in g.files if "__" in k})
    ref = net(t)  # CPU tensors -> torch-op path
    gr, gtr = torch.randn(J * 4), torch.randn(3)
    (ref["rotation"] * gr).sum().add((ref["translation"] * gtr).sum()).backward()
    net_g = copy.deepcopy(net).cuda()
    for p in net_g.parameters():
        p.grad = None
    out = net_g(t.cuda())
    assert out["rotation"].grad_fn is not None and "PoseMLPFn" in type(out["rotation"].grad_fn).__name__
    (out["rotation"] * gr.cuda()).sum().add((out["translation"] * gtr.cuda()).sum()).backward()
    U.assert_close(out["rotation"].detach().cpu().numpy(), ref["rotation"].detach().numpy(), "rotation", 1e-5)
    U.assert_close(out["translation"].detach().cpu().numpy(), ref["translation"].detach().numpy(), "translation", 1e-5)
    if width == 32:
        U.assert_close(out["rotation"].detach().cpu().numpy(), g["rotation"], "rotation vs reference", 1e-5)
    for (n, p), q in zip(net.named_parameters(), net_g.parameters()):
        U.assert_close(q.grad.cpu().numpy(), p.grad.numpy(), "grad " + n, 1e-4)


def test_pose_mlp_status_word_is_clean_after_normal_steps_and_raises_when_set():
    """The one-launch PoseMLP kernels hand data between workgroups by bounded spinning; a time-out poisons the pose with
    NaN and sets a STICKY status word that PoseMLP.check_status() / GraphedFrame.check() turn into an exception."""
    from riggs_amd import _lib as L
    sc = synth.make_scene(64, 24, 5)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8, use_skinning_weight_mlp=False,
                      use_template_offsets=False).cuda()
    x = sc["xyz"].cuda()
    for _ in range(3):
        out = sw(x, sw.expand_time(torch.tensor([0.4], device="cuda")), motion_mask=None)
        (out["d_xyz"].sum() + out["d_rotation"].sum()).backward()
    assert torch.isfinite(out["d_xyz"]).all()
    sw.pose_net.check_status()  # clean
    w = int(L.lib().riggs_pose_mlp_status_word(len(sw.pose_net.net), sw.pose_net.net[0].out_features))
    sw.pose_net._hip_sync[w] = 1  # what a timed-out spin leaves behind
    with pytest.raises(L.RiggsHipError, match="timed out"):
        sw.pose_net.check_status()
    sw.pose_net.check_status()  # cleared by the raise


def test_a_lost_handoff_surfaces_through_check_before_anything_is_exchanged_or_stepped():
    """The one-launch PoseMLP kernels hand data between workgroups with bounded spins.  When a workgroup's hand-off never
    arrives — on a GPU shared with long kernels of another stream, e.g. the collectives of an overlapped exchange, a workgroup
    may not become resident in time — the spin times out: the pose is poisoned with NaN and a sticky status word is set.
    Forced here through the library's test hook (one workgroup keeps a layer's hand-off to itself), in the forward and in the
    backward launch of a frame captured as two graphs the way the overlapped exchanges use it: GraphedFrame.check() raises
    BEFORE the caller exchanges or steps, the step is discarded, and the next replay is clean and equal to the first one.
    Then the same replay with a long kernel running on a side stream during the deformation backward: either it completes with
    the same gradients, or it is reported the same way — never silently wrong."""
    import bench
    from riggs_amd import _lib as L
    from riggs_amd.graph import GraphedFrame
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=20000, J=24, H=128, W=128)
    try:
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)
    params = bench.params_of(gm, sw)
    gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device="cuda"), params, split_backward=True).capture()
    gf.set_inputs(gimg=torch.rand(3, 128, 128, generator=torch.Generator().manual_seed(5)).cuda() / (3 * 128 * 128))

    def replay():
        gf.run_a()
        gf.run_b()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in params]

    ref = replay()
    gf.check()
    assert all(torch.isfinite(g).all() for g in ref)
    pn = sw.pose_net
    word = int(L.lib().riggs_pose_mlp_status_word(len(pn.net), pn.net[0].out_features))
    for bit in (1, 2):                       # forward launch, backward launch
        pn._hip_sync[word + 1] = bit
        bad = replay()                       # (returns after the bounded spins: a fraction of a second)
        pn._hip_sync[word + 1] = 0
        with pytest.raises(L.RiggsHipError, match="PoseMLP"):
            gf.check()                       # <- the caller's gate in front of the exchange / the optimizer
        assert not all(torch.isfinite(g).all() for g in bad)   # that step's skeleton gradients are NaN: it must be dropped
        again = replay()
        gf.check()                           # the status was cleared by the raise; this replay is clean ...
        for a, b in zip(again, ref):
            assert float((a - b).abs().max()) <= 1e-5 * max(1e-12, float(b.abs().max()))   # ... and equal to the first one
    # a long kernel on a side stream while graph (b) — FK backward, PoseMLP backward — runs
    side = torch.cuda.Stream()
    big = torch.randn(8192, 8192, device="cuda")
    gf.run_a()
    with torch.cuda.stream(side):
        for _ in range(6):
            big = torch.tanh(big @ big * 1e-4)
    gf.run_b()
    torch.cuda.synchronize()
    try:
        gf.check()
        for p, b in zip(params, ref):
            assert float((p.grad - b).abs().max()) <= 1e-5 * max(1e-12, float(b.abs().max()))
    except L.RiggsHipError:
        pass  # reported: the caller drops the step


@pytest.mark.gpu
@pytest.mark.parametrize("J,K,heads", [(24, -1, False), (64, -1, False), (12, 3, False), (24, -1, True)])
def test_forward_as_one_node_matches_pose_net_plus_deform_by_pose(J, K, heads):
    """SkeletonWarp.forward runs PoseMLP, FK + skinning (one launch) and, backward, skinning, then FK^T + PoseMLP (one launch) as
    ONE autograd node; get_pose_info + deform_by_pose are the same arithmetic over two nodes and two launches more.  Outputs
    and every gradient agree (the FK chain is evaluated by different workgroups, not in a different order)."""
    from riggs_amd import synth
    from riggs_amd.skeleton import SkeletonWarp
    sc = synth.make_scene(3000, J, 7)
    torch.manual_seed(3)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=K, hyper_dim=8, use_skinning_weight_mlp=heads,
                      use_template_offsets=heads).cuda()
    sw._node_radius.data = sc["node_radius"].cuda()
    x = sc["xyz"].cuda()
    mask = torch.rand(x.shape[0], 1, device="cuda")
    t = torch.tensor(0.41, device="cuda")
    g = torch.Generator().manual_seed(1)
    w_xyz, w_rot, w_nodes = (torch.randn(s, generator=g).cuda() for s in ((x.shape[0], 3), (x.shape[0], 4), (J, 3)))

    def run(fused):
        for p in sw.parameters():
            p.grad = None
        dv = sw(x, t, mask) if fused else sw.deform_by_pose(x, sw.get_pose_info(sw.expand_time(t)), mask)
        loss = (dv["d_xyz"] * w_xyz).sum() + (dv["d_rotation"] * w_rot).sum() + (dv["d_nodes"] * w_nodes).sum() \
            + 0.1 * (dv["local_rotation"] ** 2).sum() + 0.3 * dv["global_trans"].sum()
        loss.backward()
        return ({k: dv[k].detach().clone() for k in ("d_xyz", "d_rotation", "d_nodes", "local_rotation", "global_trans")},
                {n: p.grad.detach().clone() for n, p in sw.named_parameters() if p.grad is not None})

    o1, g1 = run(True)
    o0, g0 = run(False)
    sw.pose_net.check_status()
    for k in o0:
        assert float((o1[k] - o0[k]).abs().max()) <= 1e-6 * max(1.0, float(o0[k].abs().max())), k
    assert set(g0) == set(g1) and len(g0) >= 20
    for n in g0:
        assert float((g1[n] - g0[n]).abs().max()) <= 2e-5 * max(1e-9, float(g0[n].abs().max())), n


@pytest.mark.gpu
def test_pose_mlp_chain_placement_does_not_change_results():
    """riggs_pose_mlp_set_placement: the chain's workgroups on one XCD (plain-store hand-offs through its L2, after the in-kernel
    placement check) or spread over the device (write-through granules; what riggs_amd.dist selects for world sizes > 1) — the
    same arithmetic, bit for bit, forward and backward."""
    from riggs_amd import _lib as L
    from riggs_amd.skeleton import PoseMLP
    torch.manual_seed(5)
    net = PoseMLP(1, 24 * 4).cuda()
    bias = torch.tensor([1.0, 0, 0, 0], device="cuda")
    t = torch.tensor([0.63], device="cuda")
    res = []
    try:
        for one_xcd in (1, 0, 1):
            L.lib().riggs_pose_mlp_set_placement(one_xcd)
            for p in net.parameters():
                p.grad = None
            m = net(t, rot_bias=bias)
            ((m["rotation"] ** 2).sum() + m["translation"].sum()).backward()
            net.check_status()
            res.append([m["rotation"].detach().clone(), m["translation"].detach().clone()] + [p.grad.clone() for p in net.parameters()])
    finally:
        L.lib().riggs_pose_mlp_set_placement(1)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    for a, b in zip(res[0], res[2]):
        assert torch.equal(a, b)
