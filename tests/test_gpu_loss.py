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


# (the kernels tile the image 32 columns x 64 rows: sizes on, one under and one over the tile edges, a single pixel, a 5-pixel-wide strip)
@pytest.mark.parametrize("C,H,W", [(3, 800, 800), (1, 17, 5), (3, 33, 129), (4, 16, 16), (1, 1, 1), (2, 64, 32), (1, 65, 31), (1, 63, 33), (1, 130, 5)])
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


# ---- skeleton projection loss (train_rig.py:309-314) ---------------------------------------------------------------------
class _Cam:
    def __init__(self, z):
        self.world_view_transform = torch.from_numpy(np.asarray(z["world_view_transform"], np.float32)).cuda()
        self.FoVx, self.FoVy = float(z["FoVx"]), float(z["FoVy"])
        self.image_height, self.image_width = int(z["image_height"]), int(z["image_width"])
        self.K = z["K"] if np.size(z["K"]) else None
        self.thinned = torch.from_numpy(np.asarray(z["thinned"], np.float32)).cuda()


@pytest.mark.parametrize("name", ["skelproj_tree24_m700", "skelproj_chain8_m90_K"])
def test_skeleton_projection_loss_matches_reference_golden(name):
    from riggs_amd.loss import cal_skeleton_loss, sampling_steps
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    cam = _Cam(z)
    nodes = torch.from_numpy(z["d_nodes"]).cuda().requires_grad_(True)
    parents = torch.from_numpy(z["parents"]).cuda()
    assert sampling_steps(nodes, parents).shape[0] == int(z["steps"])
    loss = cal_skeleton_loss(nodes, parents, cam)
    (2.5 * loss).backward()
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * float(z["loss"])
    g = nodes.grad.cpu().numpy() / 2.5
    assert np.abs(g - z["grad_nodes"]).max() <= 1e-4 * np.abs(z["grad_nodes"]).max()
    # deterministic: no float atomics anywhere
    n2 = torch.from_numpy(z["d_nodes"]).cuda().requires_grad_(True)
    l2 = cal_skeleton_loss(n2, parents, cam)
    (2.5 * l2).backward()
    assert l2.item() == loss.item() and torch.equal(n2.grad, nodes.grad)
    # the trainer's weighted term formed inside the kernels: (loss, weight * loss), either root or both may be differentiated
    w = torch.tensor(0.37, device="cuda")
    n3 = torch.from_numpy(z["d_nodes"]).cuda().requires_grad_(True)
    l3, wl3 = cal_skeleton_loss(n3, parents, cam, weight=w)
    assert l3.item() == loss.item() and wl3.item() == pytest.approx(0.37 * loss.item(), rel=1e-6)
    wl3.backward(retain_graph=True)
    assert (n3.grad - 0.37 / 2.5 * nodes.grad).abs().max() <= 1e-5 * nodes.grad.abs().max()
    n3.grad = None
    torch.autograd.backward([l3, wl3], [torch.ones((), device="cuda")] * 2)
    assert (n3.grad - 1.37 / 2.5 * nodes.grad).abs().max() <= 1e-5 * nodes.grad.abs().max()


@pytest.mark.parametrize("J,M,num_sample,seed", [(2, 1, 512, 1), (64, 5000, 2048, 2), (24, 1500, 512, 3), (5, 3, 4, 4)])
def test_skeleton_projection_loss_against_oracle(J, M, num_sample, seed):
    """Edge sizes (one bone, one pixel, one step per bone) and a size beyond one LDS chunk of candidates in either direction."""
    from oracle import loss_ref as O
    from riggs_amd.loss import cal_skeleton_loss, sampling_steps, camera_intrinsics
    rng = np.random.default_rng(seed)
    parents = np.array([-1] + [int(rng.integers(0, i)) for i in range(1, J)], np.int32)
    nodes = np.zeros((J, 3), np.float32)
    for i in range(1, J):
        nodes[i] = nodes[parents[i]] + 0.2 * rng.standard_normal(3)
    nodes /= max(np.linalg.norm(nodes, axis=1).max(), 1e-3)
    view = np.eye(4, dtype=np.float32)
    view[3, :3] = [0.1, -0.05, 3.0]  # row-vector convention: translation in the last row
    z = dict(world_view_transform=view, FoVx=0.7, FoVy=0.55, image_height=200, image_width=260, K=np.zeros((0, 0)),
             thinned=np.stack([rng.integers(0, 200, M), rng.integers(0, 260, M)], -1).astype(np.float32))
    cam = _Cam(z)
    x = torch.from_numpy(nodes).cuda().requires_grad_(True)
    par = torch.from_numpy(parents).cuda()
    t = sampling_steps(x, par, num_sample)
    t_ref = O.sampling_steps(nodes, parents, num_sample)
    assert t.shape[0] == len(t_ref) and np.abs(t.cpu().numpy() - t_ref).max() <= 1.2e-7
    loss = cal_skeleton_loss(x, par, cam, t=t)
    loss.backward()
    intr = camera_intrinsics(cam)
    assert intr == tuple(float(v) for v in O.intrinsics(0.7, 0.55, 200, 260))
    want, g = O.skeleton_projection_loss(nodes, parents, view, *intr, z["thinned"], t=t.cpu().numpy())
    assert abs(loss.item() - want) <= 1e-4 * want
    # a nearest-neighbour choice decided differently in float32 and float64 (near-tie) moves single sign terms of size
    # 1/P or 1/M: compare in aggregate
    err = np.abs(x.grad.cpu().numpy() - g).max()
    assert err <= 2e-3 * np.abs(g).max(), (err, np.abs(g).max())


def test_skeleton_projection_loss_rejects_empty_sets_and_cpu_tensors():
    from riggs_amd import _lib as L
    from riggs_amd.loss import cal_skeleton_loss
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "skelproj_chain8_m90_K.npz"))
    cam = _Cam(z)
    nodes = torch.from_numpy(z["d_nodes"]).cuda()
    parents = torch.from_numpy(z["parents"]).cuda()
    with pytest.raises(L.RiggsHipError):
        cal_skeleton_loss(nodes.cpu(), parents, cam)
    with pytest.raises(L.RiggsHipError):
        cal_skeleton_loss(nodes, parents, cam, t=torch.zeros(0, device="cuda"))
    cam.thinned = torch.zeros(0, 2, device="cuda")
    with pytest.raises(L.RiggsHipError):
        cal_skeleton_loss(nodes, parents, cam)


def test_projection_loss_weights_follow_the_trainer():
    from riggs_amd.loss import ProjectionLossWeights
    w = ProjectionLossWeights(5, 1e-3)
    assert float(w.update(2, 10.0)) == pytest.approx(1e-3 * np.exp(-100.0 / (2 * (1.0e5 / 2) ** 2)))
    for uid, v in enumerate([8.0, 9.0, 10.0, 30.0, 11.0]):
        wt = w.update(uid, v)
    sigma = 10.0 / 2
    assert float(wt) == pytest.approx(1e-3 * np.exp(-11.0 ** 2 / (2 * sigma ** 2)), rel=1e-5)


def _torch_projection_loss(d_nodes, parents, cam, t):
    """The same loss written with differentiable torch ops (test-side check of the gradient hand-over into the FK backward)."""
    from riggs_amd.loss import camera_intrinsics
    fx, fy, cx, cy = camera_intrinsics(cam)
    par = parents[1:].long()
    pts = (t[:, None, None] * d_nodes[1:] + (1 - t[:, None, None]) * d_nodes[par]).reshape(-1, 3)
    V = cam.world_view_transform
    tr = pts @ V[:3, :3] + V[3, :3]
    proj = torch.stack([fy * tr[:, 1] / tr[:, 2] + cy, fx * tr[:, 0] / tr[:, 2] + cx], -1)
    d = (proj[:, None, :] - cam.thinned[None, :, :]).abs().sum(-1)
    return d.min(1).values.mean() + d.min(0).values.mean()


def test_projection_loss_gradient_reaches_the_pose_network():
    from riggs_amd import synth
    from riggs_amd.loss import cal_skeleton_loss, sampling_steps
    from riggs_amd.skeleton import SkeletonWarp
    sc = synth.make_scene(2_000, 24, 5)
    cam = synth.look_at_camera(120, 160, fid=0.4).to("cuda")
    torch.manual_seed(3)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).cuda()
    cam.thinned = torch.stack([torch.randint(20, 100, (400,)), torch.randint(30, 130, (400,))], -1).float().cuda()
    x = sc["xyz"].cuda()
    grads = {}
    for which in ("hip", "torch"):
        for p in sw.parameters():
            p.grad = None
        dv = sw(x, sw.expand_time(cam.fid), motion_mask=None)
        t = sampling_steps(dv["d_nodes"], sw.parents)
        if which == "hip":
            loss = cal_skeleton_loss(dv["d_nodes"], sw.parents, cam, t=t)
        else:
            loss = _torch_projection_loss(dv["d_nodes"], sw.parents, cam, t)
        loss.backward()
        grads[which] = (loss.item(), {n: p.grad.clone() for n, p in sw.named_parameters() if p.grad is not None})
    assert abs(grads["hip"][0] - grads["torch"][0]) <= 1e-5 * grads["torch"][0]
    assert grads["hip"][1].keys() == grads["torch"][1].keys() and len(grads["hip"][1]) >= 4
    for n, g in grads["torch"][1].items():
        assert (grads["hip"][1][n] - g).abs().max() <= 1e-3 * g.abs().max() + 1e-9, n


def test_projection_loss_inside_a_captured_training_iteration():
    import bench
    from riggs_amd import synth
    from riggs_amd.gaussian_model import GaussianModel
    from riggs_amd.graph import GraphedTrainStep
    from riggs_amd.loss import cal_skeleton_loss
    from riggs_amd.optim import FusedAdam
    from riggs_amd.skeleton import SkeletonWarp
    sc = synth.make_scene(3_000, 8, 6)
    cam = synth.look_at_camera(64, 80, fid=0.3).to("cuda")
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                    sc["opacity"], device="cuda")
    torch.manual_seed(1)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).cuda()
    gm.training_setup(bench._train_args(), capturable=True)
    opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()],
                    lr=0.0, eps=1e-15, capturable=True)
    thinned = torch.stack([torch.randint(10, 54, (150,)), torch.randint(10, 70, (150,))], -1).float().cuda()
    gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device="cuda"), torch.rand(3, 64, 80, device="cuda"), [gm.optimizer, opt],
                           thinned=thinned, projection_weight=1e-3, max_pixels=400)
    gts.capture(warmup=1)
    gts.run()
    rnd = lambda m: torch.stack([torch.randint(0, 64, (m,)), torch.randint(0, 80, (m,))], -1).float().cuda()  # noqa: E731
    for new_pixels in (None, rnd(150), rnd(400), rnd(1), rnd(333)):  # one graph, frames of any pixel count up to the capacity
        with torch.no_grad():  # the loss the NEXT replay will see: parameters as they are now
            cam.thinned = thinned if new_pixels is None else new_pixels
            cam.K = None
            d_nodes = sw(gm.get_xyz.detach()[:1], sw.expand_time(cam.fid), motion_mask=None)["d_nodes"]
            want = cal_skeleton_loss(d_nodes, sw.parents, cam, t=gts.proj_steps).item()
        out = gts.run(thinned=new_pixels)
        assert out["projection_loss"].item() == pytest.approx(want, rel=1e-6)
        if new_pixels is not None:
            thinned = new_pixels
    with pytest.raises(ValueError):
        gts.run(thinned=rnd(401))
