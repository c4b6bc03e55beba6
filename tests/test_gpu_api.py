"""-m gpu: the drop-in Python surface (render(), diff_gaussian_rasterization, simple_knn) end to end."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import deform_ref as O  # noqa: E402
from oracle import raster_ref as RR  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.gaussian_model import GaussianModel  # noqa: E402
from riggs_amd.render import render  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402
from tests import gpu_util as U  # noqa: E402


class Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = True


def _oracle_pipeline(sc, cam, bg, gimg, isotropic=False):
    """deform (torch CPU oracle) -> glue -> C rasterizer fwd+bwd -> autograd back to the raw parameters."""
    leaf = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    P = {k: leaf(sc[k]) for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity",
                                  "local_rotation", "global_trans", "node_radius")}
    scaling = P["scaling"][:, :1] if isotropic else P["scaling"]
    dv = O.deform_by_pose(P["xyz"].detach(), sc["joints"], sc["parents"], P["node_radius"], P["local_rotation"],
                          P["global_trans"], sc["motion_mask"], -1)
    d_rot = dv["d_rotation"] * (0.0 if isotropic else 1.0)
    m3, op, scl, rot, shs = O.render_glue(P["xyz"], P["features_dc"], P["features_rest"], scaling, P["rotation"],
                                          P["opacity"], dv["d_xyz"], d_rot, dv["d_scaling"], isotropic)
    out, saved = RR.forward(m3.detach().numpy(), op.detach().numpy(), cam.world_view_transform.numpy(),
                            cam.full_proj_transform.numpy(), cam.camera_center.numpy(), math.tan(cam.FoVx / 2),
                            math.tan(cam.FoVy / 2), cam.image_height, cam.image_width, np.asarray(bg, np.float32),
                            shs=shs.detach().numpy(), scales=scl.detach().numpy(), rotations=rot.detach().numpy())
    g = RR.backward(saved, gimg.numpy(), None, None)
    T = torch.from_numpy
    torch.autograd.backward([m3, op, scl, rot, shs], [T(g["means3D"]), T(g["opacities"]), T(g["scales"]),
                                                      T(g["rotations"]), T(g["shs"])])
    return out, saved, P, g


@pytest.mark.parametrize("fused,isotropic", [(True, False), (False, False), (True, True)])
def test_render_end_to_end_matches_oracle_pipeline(fused, isotropic):
    N, J, H, W = 6000, 24, 160, 160
    sc = synth.make_scene(N, J, 77, scale=0.02)
    if isotropic:
        sc["scaling"] = sc["scaling"][:, :1].contiguous()
    cam = synth.look_at_camera(H, W)
    bg = [0.0, 0.0, 0.0]
    g = torch.Generator().manual_seed(4)
    gimg = torch.sign(torch.rand(3, H, W, generator=g) - 0.5) / (3 * H * W)
    out_o, saved_o, Po, g_o = _oracle_pipeline(sc, cam, bg, gimg, isotropic)

    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                    sc["opacity"], use_isotropic_gs=isotropic)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8,
                      use_skinning_weight_mlp=False, use_template_offsets=False).cuda()
    sw._node_radius.data = sc["node_radius"].cuda()
    q = sc["local_rotation"].cuda().requires_grad_(True)
    gt = sc["global_trans"].cuda().requires_grad_(True)
    dv = sw.deform_by_pose(gm.get_xyz.detach(), {"local_rotation": q, "global_trans": gt}, gm.motion_mask)
    d_rot = dv["d_rotation"] * 0.0 if isotropic else dv["d_rotation"]  # train_rig.py:422-423
    pkg = render(cam.to("cuda"), gm, Pipe, torch.tensor(bg, device="cuda"), dv["d_xyz"], d_rot, dv["d_scaling"],
                 fused=fused)
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii", "depth", "alpha", "bg_color"}
    # the deformed means differ from the oracle's by float rounding, so ordering is compared with tolerance here
    U.assert_close(pkg["render"].detach().cpu().numpy(), out_o["color"], "render", U.REL_TOL, 1e-3)
    assert (pkg["radii"].cpu().numpy() != saved_o.radii).mean() < 1e-3
    assert torch.equal(pkg["visibility_filter"], pkg["radii"] > 0)
    (pkg["render"] * gimg.cuda()).sum().backward()
    vs = pkg["viewspace_points"].grad
    U.assert_close(vs.cpu().numpy(), g_o["means2D"], "viewspace_points.grad", 2e-4, 1e-3)
    for name, p in (("xyz", gm._xyz), ("features_dc", gm._features_dc), ("features_rest", gm._features_rest),
                    ("opacity", gm._opacity), ("scaling", gm._scaling), ("rotation", gm._rotation)):
        if isotropic and name == "rotation":
            # Sigma = s^2 R R^T = s^2 I: the rotation gradient is identically zero in exact arithmetic and pure
            # rounding noise in both implementations — check it is negligible instead of comparing noise.
            assert float(p.grad.abs().max()) < 1e-3 * float(gm._xyz.grad.abs().max())
            continue
        U.assert_close(p.grad.cpu().numpy(), Po[name].grad.numpy(), "dL/d_" + name, 2e-4, 1e-3)
    U.assert_close(q.grad.cpu().numpy(), Po["local_rotation"].grad.numpy(), "dL/dlocal_rotation", 1e-3)
    U.assert_close(gt.grad.cpu().numpy(), Po["global_trans"].grad.numpy(), "dL/dglobal_trans", 1e-3)
    U.assert_close(sw._node_radius.grad.cpu().numpy(), Po["node_radius"].grad.numpy(), "dL/d_node_radius", 1e-3)


def test_render_accepts_python_float_residuals_and_override_color():
    sc = synth.make_scene(2000, 8, 3, scale=0.03)
    cam = synth.look_at_camera(64, 64).to("cuda")
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                    sc["opacity"])
    bg = torch.zeros(3, device="cuda")
    a = render(cam, gm, Pipe, bg, 0.0, 0.0, 0.0, fused=True)       # train_gui.py:1032 passes floats
    b = render(cam, gm, Pipe, bg, 0.0, 0.0, 0.0, fused=False)
    U.assert_close(a["render"].detach().cpu().numpy(), b["render"].detach().cpu().numpy(), "fused vs general", 1e-5, 1e-4)
    col = torch.rand(2000, 3, device="cuda")
    c = render(cam, gm, Pipe, bg, 0.0, 0.0, 0.0, override_color=col)
    assert c["render"].shape == (3, 64, 64) and torch.isfinite(c["render"]).all()


def test_drop_in_module_surfaces():
    import diff_gaussian_rasterization as dgr
    from simple_knn._C import distCUDA2
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    assert issubclass(dgr.GaussianRasterizer, torch.nn.Module)
    g = torch.Generator().manual_seed(0)
    for P in (1, 3, 24, 5000):
        pts = torch.randn(P, 3, generator=g)
        ref = RR.dist2_knn3(pts.numpy())
        got = distCUDA2(pts.cuda()).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-9)


def test_flat_gradient_bucket_receives_gradients_in_place():
    """riggs_amd.dist: with a registered bucket every HIP backward writes dL/dparam into its slice of ONE flat buffer
    (p.grad aliases it, the all-reduce needs no pack), and the values equal the unregistered run's."""
    import bench
    from riggs_amd.dist import FlatGradAllReduce
    from riggs_amd.rasterizer import RasterArena
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=6001, J=8, H=96, W=112)
    try:
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
        params = bench.params_of(gm, sw)
        gimg = torch.rand(3, 96, 112, generator=torch.Generator().manual_seed(1)).cuda()

        def run(bucket):
            step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, bucket)
            step()
            torch.cuda.synchronize()
            return [p.grad.clone() for p in params]
        plain = FlatGradAllReduce(params, register=False)
        ref = run(plain)
        assert all(p.grad.data_ptr() != v.data_ptr() for p, v in zip(params, plain.views))
        bucket = FlatGradAllReduce(params)
        got = run(bucket)
        lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + bucket.flat.numel() * 4
        assert all(lo <= p.grad.data_ptr() < hi and p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, bucket.views))
        assert all(v.data_ptr() % 16 == 0 for v in bucket.views)
        for a, b in zip(got, ref):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-6 * float(b.abs().max()) + 1e-12)  # float atomics: order varies
        bucket()  # world 1: no collective, gradients stay in place
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, bucket.views))
        bucket.unregister()
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)


@pytest.mark.parametrize("sparse_rows", [False, True])
def test_graphed_train_step_matches_eager_iterations(sparse_rows):
    """A whole training iteration (deform, render, fused loss, backward, capturable FusedAdam on Gaussians AND skeleton)
    captured once and replayed, against the same iterations issued eagerly with torch.optim.Adam and the torch-op loss glue."""
    import copy
    from types import SimpleNamespace

    import bench
    from riggs_amd.graph import GraphedTrainStep
    from riggs_amd.loss import l1_loss, ssim
    from riggs_amd.optim import FusedAdam
    from riggs_amd.rasterizer import RasterArena
    from riggs_amd.render import render
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=5000, J=8, H=80, W=96)
    args = SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                           position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.001, rotation_lr=0.001)
    try:
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
        gm2, sw2 = copy.deepcopy(gm), copy.deepcopy(sw)
        gt = torch.rand(3, 80, 96, generator=torch.Generator().manual_seed(2)).cuda()
        bg = torch.zeros(3, device="cuda")
        # --- graph side
        gm.training_setup(args, capturable=True)
        sk_opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()],
                           lr=0.0, eps=1e-15, capturable=True)
        gts = GraphedTrainStep(gm, sw, cam, bg, gt, [gm.optimizer, sk_opt], lambda_dssim=0.2, sparse_grad_rows=sparse_rows)
        gts.capture(warmup=1)          # one eager warm-up iteration (iteration 1)
        assert bool(gts.sparse_outputs) == sparse_rows
        losses = []
        for it in range(2, 5):         # iterations 2..4 are replays; xyz learning rate rescheduled in between
            gm.update_learning_rate(1000 * it)
            out = gts.run()
            losses.append(out["loss"].item())
        gts.check()
        # --- eager reference side: torch.optim.Adam, loss glue in torch ops
        gm2.training_setup(args)
        opt_g = torch.optim.Adam([{"params": g["params"], "lr": float(g["lr"]), "name": g["name"]} for g in gm2.optimizer.param_groups],
                                 lr=0.0, eps=1e-15)
        opt_s = torch.optim.Adam([{"params": g["params"], "lr": 5e-4} for g in sw2.trainable_parameters()], lr=0.0, eps=1e-15)
        ref_losses = []
        for it in range(1, 5):
            if it >= 2:
                for grp in opt_g.param_groups:
                    if grp["name"] == "xyz":
                        grp["lr"] = gm2.xyz_scheduler_args(1000 * it)
            opt_g.zero_grad(set_to_none=True), opt_s.zero_grad(set_to_none=True)
            dv = sw2(gm2.get_xyz.detach(), sw2.expand_time(cam.fid), motion_mask=gm2.motion_mask)
            pkg = render(cam, gm2, bench.Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"], arena=RasterArena())
            loss = 0.8 * l1_loss(pkg["render"], gt) + 0.2 * (1.0 - ssim(pkg["render"], gt))
            loss.backward()
            opt_g.step(), opt_s.step()
            if it >= 2:
                ref_losses.append(loss.item())
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-4)
        assert losses[-1] < losses[0]  # it trains
        for a, b in zip(gm.parameters(), gm2.parameters()):
            # (as below: an element whose gradient is at the float-atomics noise level may step the other way)
            a, b = a.detach(), b.detach()
            bad = (a - b).abs() > 2e-3 * b.abs() + 2e-4 * float(b.abs().max())
            assert float(bad.float().mean()) <= 1e-3, float(bad.float().mean())
        for a, b in zip(sw.pose_net.parameters(), sw2.pose_net.parameters()):
            # Adam moves an element by ~lr per step whatever the size of its gradient: where the gradient is at the level of
            # the float-atomics noise of the compositing backward its SIGN differs between two runs, so a handful of the
            # 0.6 M PoseMLP weights may differ by up to 2 x 4 steps x lr; everything else must agree closely
            a, b = a.detach(), b.detach()
            tol = 2e-3 * b.abs() + 2e-4 * float(b.abs().max()) + 1e-7
            bad = (a - b).abs() > tol
            assert float(bad.float().mean()) <= 1e-3, float(bad.float().mean())
            assert float((a - b).abs().max()) <= 8 * 5e-4 + 1e-6
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)


def test_capture_of_a_drifting_scene_records_no_arena_allocation():
    """Training iterations change the instance count from frame to frame.  The arena keeps its allocation while the count stays
    inside a third of the head-room, and the owner of a capture tops the head-room up BEFORE capturing: the captured iteration
    must contain neither an allocation of the arena nor the memset that clears its walk history (they would replay for ever:
    9 us of every iteration, and the forward's wide blocks never engaged — seen in a kernel trace of the captured iteration)."""
    from types import SimpleNamespace

    import bench
    from riggs_amd import _lib as L
    from riggs_amd.graph import GraphedTrainStep
    from riggs_amd.optim import FusedAdam
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=20000, J=8, H=160, W=160)
    args = SimpleNamespace(percent_dense=0.01, position_lr_init=0.016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                           position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.05, rotation_lr=0.001)
    lib = L.lib()
    orig = lib.riggs_raster_binning_reset_history
    calls = []

    def spy(*a):
        calls.append(bool(torch.cuda.is_current_stream_capturing()))
        return orig(*a)
    try:
        lib.riggs_raster_binning_reset_history = spy
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
        gt = torch.rand(3, 160, 160, generator=torch.Generator().manual_seed(2)).cuda()
        gm.training_setup(args, capturable=True)  # (large position / scale steps: the count moves by per cents per iteration)
        sk_opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()],
                           lr=0.0, eps=1e-15, capturable=True)
        gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device="cuda"), gt, [gm.optimizer, sk_opt], sparse_grad_rows=True)
        gts.capture(warmup=3)
        assert calls and not any(calls), calls   # (resets happened, all of them eagerly)
        where, cap = gts.arena.binning.data_ptr(), gts.arena.capacity
        counts = []
        for _ in range(4):
            gts.run()
            counts.append(gts.check())
        assert gts.arena.binning.data_ptr() == where and gts.arena.capacity == cap
        assert len(set(counts)) > 1, counts     # the scene did drift
        assert cap >= int(min(counts) * 1.3)    # ... inside the topped-up head-room (1.5)
    finally:
        lib.riggs_raster_binning_reset_history = orig
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)


def test_graphed_train_step_survives_densification_by_recapture():
    """The reference changes N every densification_interval iterations (scene/gaussian_model.py:445-514: clone / split /
    prune, each replacing the parameter tensors and moving the optimizer state with cat_tensors_to_optimizer /
    _prune_optimizer).  A captured iteration has N, the arena capacity and the gradient buffers baked in:
    GraphedTrainStep.recapture() drops them and captures again over the new tensors.  Here: two replays, a densify-style
    change of N (clone 400 Gaussians, prune 300, optimizer state moved the way the reference's surgery moves it), recapture,
    two more iterations — against the same iterations issued eagerly with torch.optim.Adam from the same state."""
    import copy
    from types import SimpleNamespace

    import bench
    from riggs_amd.graph import GraphedTrainStep
    from riggs_amd.loss import l1_loss, ssim
    from riggs_amd.optim import FusedAdam
    from riggs_amd.rasterizer import RasterArena
    from riggs_amd.render import render
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=5000, J=8, H=80, W=96)
    args = SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                           position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.001, rotation_lr=0.001)
    names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
             "rotation": "_rotation"}

    def surgery(gm_, keep_idx):
        """new tensor = old[keep_idx] for every Gaussian parameter; optimizer state follows (the reference's
        cat_tensors_to_optimizer + _prune_optimizer in one index_select)"""
        for group in gm_.optimizer.param_groups:
            old_p = group["params"][0]
            st = gm_.optimizer.state.pop(old_p, None)
            new_p = torch.nn.Parameter(old_p.detach()[keep_idx].contiguous().requires_grad_(True))
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][keep_idx].contiguous(), st["exp_avg_sq"][keep_idx].contiguous()
                gm_.optimizer.state[new_p] = st
            group["params"][0] = new_p
            setattr(gm_, names[group["name"]], new_p)
        n = keep_idx.numel()
        gm_.xyz_gradient_accum = torch.zeros((n, 1), device="cuda")
        gm_.denom = torch.zeros((n, 1), device="cuda")
        gm_.max_radii2D = torch.zeros(n, device="cuda")
    try:
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
        gt = torch.rand(3, 80, 96, generator=torch.Generator().manual_seed(2)).cuda()
        bg = torch.zeros(3, device="cuda")
        gm.training_setup(args, capturable=True)
        sk_opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()],
                           lr=0.0, eps=1e-15, capturable=True)
        gts = GraphedTrainStep(gm, sw, cam, bg, gt, [gm.optimizer, sk_opt], lambda_dssim=0.2, sparse_grad_rows=True)
        gts.capture(warmup=1)
        gts.run(), gts.run()
        gts.check()
        torch.cuda.synchronize()
        # ---- densify-style change of N: clone the first 400, prune the last 300
        N0 = gm.get_xyz.shape[0]
        keep = torch.cat([torch.arange(N0 - 300), torch.arange(400)]).cuda()
        surgery(gm, keep)
        N1 = gm.get_xyz.shape[0]
        assert N1 == N0 + 100
        # the eager reference starts from the very same state
        gm2, sw2 = copy.deepcopy(gm), copy.deepcopy(sw)
        gm2.optimizer = None
        opt_g = torch.optim.Adam([{"params": [getattr(gm2, names[g["name"]])], "lr": float(g["lr"]), "name": g["name"]}
                                  for g in gm.optimizer.param_groups], lr=0.0, eps=1e-15)
        for g, g2 in zip(gm.optimizer.param_groups, opt_g.param_groups):
            st = gm.optimizer.state[g["params"][0]]
            opt_g.state[g2["params"][0]] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"].clone(),
                                            "exp_avg_sq": st["exp_avg_sq"].clone()}
        opt_s = torch.optim.Adam([{"params": g["params"], "lr": 5e-4} for g in sw2.trainable_parameters()], lr=0.0, eps=1e-15)
        for (p, p2) in zip([q for g in sw.trainable_parameters() for q in g["params"]], [q for g in sw2.trainable_parameters() for q in g["params"]]):
            st = sk_opt.state[p]
            opt_s.state[p2] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone()}
        # ---- graph side: recapture (its one eager warm-up frame is iteration A), then replays B and C
        gts.recapture(warmup=1)
        assert gts.out["radii"].shape[0] == N1 and bool(gts.sparse_outputs)
        losses = [gts.run()["loss"].item(), gts.run()["loss"].item()]
        gts.check()
        # ---- eager side: iterations A, B, C
        ref_losses = []
        for it in range(3):
            opt_g.zero_grad(set_to_none=True), opt_s.zero_grad(set_to_none=True)
            dv = sw2(gm2.get_xyz.detach(), sw2.expand_time(cam.fid), motion_mask=gm2.motion_mask)
            pkg = render(cam, gm2, bench.Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"], arena=RasterArena())
            loss = 0.8 * l1_loss(pkg["render"], gt) + 0.2 * (1.0 - ssim(pkg["render"], gt))
            loss.backward()
            opt_g.step(), opt_s.step()
            ref_losses.append(loss.item())
        np.testing.assert_allclose(losses, ref_losses[1:], rtol=2e-4)
        for a, b in zip(gm.parameters(), gm2.parameters()):
            a, b = a.detach(), b.detach()
            assert a.shape == b.shape and a.shape[0] == N1
            bad = (a - b).abs() > 2e-3 * b.abs() + 2e-4 * float(b.abs().max())
            assert float(bad.float().mean()) <= 1e-3, float(bad.float().mean())
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)


@pytest.mark.parametrize("with_mask,extra_heads", [(False, False), (True, True)])
def test_frame_entry_equals_the_two_calls_and_the_oracle(with_mask, extra_heads):
    """riggs_amd.frame.deform_render — the frame as ONE autograd node over riggs_frame_forward / riggs_frame_backward — against
    the two calls it replaces (SkeletonWarp.forward + render(fused=True): the same kernels, so the image is bitwise the same and
    the gradients agree to the float atomics' reordering) and against the CPU oracle; with cotangents on d_nodes /
    local_rotation / global_trans (the projection loss and the pose regularisers of train_rig.py hang on them) and a motion
    mask.  The first frame of an arena goes through the separate calls (its size is not known yet): both kinds are covered."""
    from riggs_amd.frame import deform_render
    from riggs_amd.rasterizer import RasterArena
    N, J, H, W = 6000, 24, 128, 160
    sc = synth.make_scene(N, J, 1240, scale=0.03)
    cam = synth.look_at_camera(H, W, fid=0.41).to("cuda")
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"], sc["opacity"])
    if with_mask:
        gm.with_motion_mask = True
        gm.feature = torch.nn.Parameter(torch.randn(N, 9, generator=torch.Generator().manual_seed(2)).cuda())
    torch.manual_seed(3)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8, use_skinning_weight_mlp=False,
                      use_template_offsets=False).cuda()
    sw._node_radius.data = sc["node_radius"].cuda()
    with torch.no_grad():
        sw.pose_net.rotation_predictor.weight.mul_(0.1)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    g = torch.Generator().manual_seed(4)
    gimg = (torch.sign(torch.rand(3, H, W, generator=g) - 0.5) / (3 * H * W)).cuda()
    gn, gq, gt = (torch.randn(J, 3, generator=g) * 1e-3).cuda(), (torch.randn(J, 4, generator=g) * 1e-3).cuda(), (torch.randn(3, generator=g) * 1e-3).cuda()
    params = gm.parameters() + [sw._node_radius] + list(sw.pose_net.parameters()) + ([gm.feature] if with_mask else [])

    def run(frame_entry, arena):
        for p in params:
            p.grad = None
        if frame_entry:
            pkg = deform_render(cam, gm, sw, Pipe, bg, arena=arena)
            dn, lq, tr = pkg["d_nodes"], pkg["local_rotation"], pkg["global_trans"]
        else:
            dv = sw(gm.get_xyz.detach(), sw.expand_time(cam.fid), motion_mask=gm.motion_mask)
            pkg = render(cam, gm, Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"], fused=True, arena=arena)
            dn, lq, tr = dv["d_nodes"], dv["local_rotation"], dv["global_trans"]
        loss = (pkg["render"] * gimg).sum()
        if extra_heads:
            loss = loss + (dn * gn).sum() + (lq * gq).sum() + (tr.reshape(-1) * gt).sum()
        loss.backward()
        torch.cuda.synchronize()
        return pkg["render"].detach().clone(), pkg["radii"].clone(), [p.grad.detach().clone() for p in params], \
            pkg["viewspace_points"].grad.detach().clone()
    a_ref = RasterArena(min_capacity=16)
    run(False, a_ref)
    img0, radii0, grads0, vg0 = run(False, a_ref)
    a_new = RasterArena(min_capacity=16)
    img_first, _, grads_first, _ = run(True, a_new)      # first frame of the arena: the separate calls
    assert a_new.last_R >= 0
    img1, radii1, grads1, vg1 = run(True, a_new)          # the frame entry
    img2, _, grads2, _ = run(True, a_new)
    assert torch.equal(img0, img1) and torch.equal(img0, img_first) and torch.equal(img1, img2) and torch.equal(radii0, radii1)
    for k, (a, b) in enumerate(zip(grads0 + [vg0], grads1 + [vg1])):
        scale = float(a.abs().max())
        assert scale > 0, k
        assert float((a - b).abs().max()) <= 2e-5 * scale, (k, float((a - b).abs().max()) / scale)
    # against the oracle: image
    o = O.deform_by_pose(sc["xyz"], sc["joints"], sc["parents"], sc["node_radius"], None, None, None, -1) if False else None
    with torch.no_grad():
        pose = sw.get_pose_info(sw.expand_time(cam.fid))
    mask_cpu = gm.motion_mask.detach().cpu() if with_mask else sc["motion_mask"]
    o = O.deform_by_pose(sc["xyz"], sc["joints"], sc["parents"], sc["node_radius"], pose["local_rotation"].cpu(), pose["global_trans"].cpu(),
                         mask_cpu, -1)
    m3, op, scl, rot, shs = O.render_glue(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"], sc["opacity"],
                                          o["d_xyz"], o["d_rotation"], o["d_scaling"])
    camc = cam.to("cpu")
    out_o, _ = RR.forward(m3.numpy(), op.numpy(), camc.world_view_transform.numpy(), camc.full_proj_transform.numpy(),
                          camc.camera_center.numpy(), math.tan(camc.FoVx / 2), math.tan(camc.FoVy / 2), H, W, bg.cpu().numpy(),
                          shs=shs.numpy(), scales=scl.numpy(), rotations=rot.numpy())
    U.assert_close(img1.cpu().numpy(), out_o["color"], "frame entry image vs oracle", U.REL_TOL, 1e-3)


def test_bench_contract_single_and_two_ranks():
    """bench.py end to end on a tiny scene: the N = 1 JSON line carries every field of the contract (roofline, cpu_baseline,
    train_step), and the 2-rank launch (the driver's torch.distributed.run command line; gloo stands in for RCCL on a
    1-GPU box) completes — no collective is ever issued by a subset of the ranks — and reports the whole-job aggregate."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RIGGS_BENCH_TEST_WORKLOAD="4000,8,96,112", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2"], cwd=root, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["scaling"] == "weak" and d["dtype"] == "f32" and d["value"] > 0
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["peak"] == 8000.0
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert d["train_step"]["value"] > 0
    assert d["cpu_baseline"]["c1_10k_chain8_256"]["ms_median"] > 0 and d["cpu_baseline"]["cpu_model"]
    # the timed frame was checked against the CPU oracle at the bench size (bench.py exits non-zero beyond the bar)
    assert d["parity_at_bench_size"]["worst_outlier_frac"] <= 5e-5 and "image" in d["parity_at_bench_size"]
    assert d["parity_at_bench_size"]["dL/d_pose_net"]["max_rel"] <= 1e-4 and d["parity_at_bench_size"]["worst_small_tensor_max_rel"] <= 1e-4
    assert d["roofline"]["bound"] in ("valu", "hbm") and 0 < d["dense_gradient_scene"]["gaussians_with_gradient"] <= 1
    # ... after the HIP deformation's own forward values were compared with the oracle's (before they replace them)
    dp = d["parity_at_bench_size"]["deform_forward_vs_oracle_max_rel"]
    assert dp["d_xyz"] <= 1e-4 and dp["d_rotation"] <= 1e-4
    # the data-parallel step's sequence on a one-rank RCCL communicator (a child process): the exchange calls issued eagerly
    # around two graphs, and the whole step — collectives included — captured as one graph
    # the eagerly issued frame, as the reference's two calls and through the frame entry
    assert 0 < d["eager_api"]["frame_entry_ms"] and 0 < d["eager_api"]["two_calls_ms"]
    ex = d["exchange_path"]
    assert "error" not in ex, ex
    assert ex["two_graphs_eager_collectives_ms"] > 0 and ex["plain_frame_ms"] > 0 and ex["rows_needed"] > 0
    # (the captured form is reported when the capture succeeded; a failure is recorded in the line, not hidden)
    assert ex.get("one_graph_ms", 0) > 0 or "one_graph_error" in ex, ex
    assert d["exchange_path_ms"] == ex.get("one_graph_ms", ex["two_graphs_eager_collectives_ms"]) == ex["ms"]
    assert ex["form"] == ("one_graph" if ex["captured_collectives_work"] else "two_graphs_eager_collectives")
    # ... and the same on the opaque-skin scene, whose segments are the large ones
    dx = ex["dense_scene"]
    assert "error" not in dx, dx
    assert dx["rows_needed"] > 0 and dx["segment_MB"] > 0 and dx["ms"] > 0 and "opaque" in dx["scene"]
    env["RIGGS_BENCH_BACKEND"] = "gloo"
    import socket

    def free_port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return str(sk.getsockname()[1])
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", free_port(), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d2 = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d2["n_gpus"] == 2 and d2["config"]["parallelism"] == "frames x2" and "cpu_baseline" not in d2
    assert abs(d2["value"] - 2 * 4 / (d2["ms_per_step"] * 4 / 1e3)) < 1e-3 * d2["value"]  # aggregate over both ranks
    # the default exchange is the packed-row one (bench.py itself compares it with a plain all-reduce after the timed region)
    assert d2["config"]["exchange"].startswith("packed rows")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", free_port(), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--exchange", "dense"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d3 = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d3["n_gpus"] == 2 and d3["config"]["exchange"].startswith("two-phase")
    # ... and exactly `python bench.py --gpus 2`: it starts its two ranks itself (what a driver that calls the scaling runs the
    # way it calls the single-GPU one does)
    env.pop("MASTER_ADDR", None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d4 = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d4["n_gpus"] == 2 and d4["steps"] == 3 and d4["config"]["parallelism"] == "frames x2" and d4["value"] > 0


def test_dist_knn3_at_initialisation_size():
    """simple_knn.distCUDA2 at the size the reference calls it with (scene/gaussian_model.py:170: up to ~150k initial points):
    exact 3-NN means against a k-d tree (scipy), and the launch is timed (exact brute force: O(P^2), tens of ms)."""
    import time
    from scipy.spatial import cKDTree
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(7)
    pts = torch.randn(150_000, 3, generator=g) * torch.tensor([1.0, 0.6, 0.3])
    d, _ = cKDTree(pts.double().numpy()).query(pts.double().numpy(), k=4)
    ref = (d[:, 1:] ** 2).mean(1)
    x = pts.cuda()
    distCUDA2(x[:1000])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = distCUDA2(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=1e-10)
    assert dt < 1.0, "distCUDA2(150k) took %.3f s" % dt
    print("distCUDA2(150k points): %.1f ms" % (dt * 1e3))


@pytest.mark.parametrize("kind,P", [("gauss", 2048), ("gauss", 60_001), ("clusters_and_outliers", 40_000), ("plane", 20_000),
                                    ("line", 5_000), ("duplicates", 9_000), ("lattice", 32_768), ("big", 2_000_000)])
def test_grid_knn_equals_the_all_pairs_search(kind, P):
    """riggs_dist2_knn3 from 2048 points on: uniform grid + ring search.  It must return what the exact all-pairs kernel
    returns (riggs_dist2_knn3_bruteforce; same distance expression: equal to rounding) on clouds that stress the grid —
    far outliers (rings widen to the whole grid), a plane and a line (degenerate axes), duplicated points (zero distances, more
    points in a cell than neighbours asked for), a lattice (points on cell faces) — and stay fast at 2 M points."""
    import time
    from riggs_amd import _lib as L
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(P)
    if kind == "gauss" or kind == "big":
        pts = torch.randn(P, 3, generator=g) * torch.tensor([1.0, 0.6, 0.3])
    elif kind == "clusters_and_outliers":
        c = torch.randn(20, 3, generator=g) * 3
        pts = c[torch.randint(0, 20, (P,), generator=g)] + 0.01 * torch.randn(P, 3, generator=g)
        pts[:7] = torch.tensor([[500.0, 0, 0], [0, -800.0, 3], [1e3, 1e3, 1e3], [-1e3, 2, 2], [0, 0, 900.0], [901.0, 0, 0], [901.0, 0.1, 0]])
    elif kind == "plane":
        pts = torch.cat([torch.rand(P, 2, generator=g), torch.full((P, 1), 0.25)], 1)
    elif kind == "line":
        pts = torch.stack([torch.rand(P, generator=g), torch.full((P,), -1.0), torch.full((P,), 2.0)], 1)
    elif kind == "duplicates":
        base = torch.randn(P // 9, 3, generator=g)
        pts = base.repeat(9, 1)[torch.randperm(P // 9 * 9, generator=g)]
    else:
        k = round(P ** (1 / 3))
        ax = torch.arange(k, dtype=torch.float32) * 0.125
        pts = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    pts = pts.contiguous()
    n = pts.shape[0]
    x = pts.cuda()
    got = distCUDA2(x)
    torch.cuda.synchronize()
    if kind == "big":
        t0 = time.perf_counter()
        got = distCUDA2(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert dt < 0.5, "distCUDA2(2M points) took %.3f s" % dt
        print("distCUDA2(2M points): %.1f ms" % (dt * 1e3))
        sub = torch.randperm(n, generator=g)[:2000]
        d = ((pts[sub, None, :].double() - pts[None, :, :].double()) ** 2).sum(-1) if False else None
        from scipy.spatial import cKDTree
        dd, _ = cKDTree(pts.double().numpy()).query(pts[sub].double().numpy(), k=4)
        np.testing.assert_allclose(got.cpu().numpy()[sub.numpy()], (dd[:, 1:] ** 2).mean(1), rtol=2e-4, atol=1e-10)
        return
    ref = torch.empty(n, device="cuda")
    L.check(L.lib().riggs_dist2_knn3_bruteforce(n, x.data_ptr(), ref.data_ptr(), L.stream_ptr()), "riggs_dist2_knn3_bruteforce")
    torch.cuda.synchronize()
    a, b = got.cpu().numpy(), ref.cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-12 * max(1.0, float(np.abs(b).max())))
    assert np.isfinite(a).all()


def test_captured_frame_skips_the_zero_fill_of_untouched_rows_without_leaving_stale_gradients():
    """riggs_raster_cfg.sparse_zero through GraphedFrame: replay after replay with changing cameras (so that rows gain and
    lose their gradient), every parameter gradient and the screen-space gradient equal those of a frame that rewrites every
    row — in particular a row that had a gradient in the previous replay and has none now is exactly zero."""
    import bench
    from riggs_amd import synth
    from riggs_amd.graph import GraphedFrame
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=6001, J=8, H=96, W=112)
    try:
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)
    import copy
    gm2, sw2 = copy.deepcopy(gm), copy.deepcopy(sw)  # (its own parameters: the two frames must not share gradient buffers)
    gimg = torch.rand(3, 96, 112, generator=torch.Generator().manual_seed(5)).cuda()
    sparse = GraphedFrame(gm, sw, cam, torch.zeros(3, device="cuda"), bench.params_of(gm, sw), sparse_grad_rows=True).capture()
    full = GraphedFrame(gm2, sw2, cam, torch.zeros(3, device="cuda"), bench.params_of(gm2, sw2), sparse_grad_rows=False).capture()
    assert sparse.sparse_rows and len(sparse.sparse_outputs) >= 6 and not full.sparse_outputs
    assert not {t.data_ptr() for t in sparse.grads} & {t.data_ptr() for t in full.grads}
    sparse.set_inputs(gimg=gimg)
    full.set_inputs(gimg=gimg)
    lost_rows = 0
    prev_nz = None
    for az, el in ((45.0, 20.0), (170.0, -35.0), (290.0, 60.0), (45.0, 20.0), (100.0, 0.0)):
        c = synth.look_at_camera(96, 112, azimuth_deg=az, elevation_deg=el, radius=3.0).to("cuda:0")
        a = sparse.run(cam=c)
        b = full.run(cam=c)
        torch.cuda.synchronize()
        pairs = list(zip(sparse.grads, full.grads)) + [(a["viewspace_points_grad"], b["viewspace_points_grad"])]
        for ga, gb in pairs:
            assert int(((ga != 0) & (gb == 0)).sum()) == 0                      # nothing stale
            torch.testing.assert_close(ga, gb, rtol=1e-4, atol=2e-6 * float(gb.abs().max()) + 1e-12)  # (float atomics: order varies)
        nz = (full.grads[0].reshape(6001, -1) != 0).any(1)
        if prev_nz is not None:
            lost_rows += int((prev_nz & ~nz).sum())
        prev_nz = nz
    assert lost_rows > 100                                                     # the sequence did exercise "had a gradient, has none now"
    # another writer (say a dense all-reduce in place) fills every row: after mark_all_rows() the next replay rewrites them all
    for t in sparse.sparse_outputs:
        t.fill_(7.0)
    sparse.mark_all_rows()
    a = sparse.run()
    b = full.run()
    torch.cuda.synchronize()
    for ga, gb in list(zip(sparse.grads, full.grads)) + [(a["viewspace_points_grad"], b["viewspace_points_grad"])]:
        assert int(((ga != 0) & (gb == 0)).sum()) == 0
        torch.testing.assert_close(ga, gb, rtol=1e-4, atol=2e-6 * float(gb.abs().max()) + 1e-12)
    sparse.reset_sparse_rows()
    assert all(float(t.abs().max()) == 0.0 for t in sparse.sparse_outputs)
