"""Generate golden vectors by importing the REAL RigGS reference (CPU) — run in the
build container only:   python tests/golden/make_golden.py

Emits small .npz fixtures (inputs + expected outputs) next to this file.  The
reference Python itself is never copied; fixtures are data.
SURVEY.md §8-c G1..G4, G7..G9.
"""
import os
import sys
import math

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim as S  # noqa: E402

S.install()
with S.quiet():
    from skeleton_utils.skeleton_warp import SkeletonWarp  # noqa: E402
    from skeleton_utils.network_utils import PoseMLP  # noqa: E402
    from utils.time_utils import quaternion_to_matrix, matrix_to_quaternion  # noqa: E402
    from utils.sh_utils import eval_sh  # noqa: E402
    from utils.general_utils import build_scaling_rotation, strip_symmetric  # noqa: E402
    from gaussian_renderer import render  # noqa: E402
    from scene.gaussian_model import GaussianModel  # noqa: E402
    from scene.cameras import Camera  # noqa: E402


def np_(t):
    return t.detach().cpu().numpy()


def random_tree(g, J, chain=False):
    if chain:
        parents = torch.arange(-1, J - 1)
        joints = torch.stack([torch.zeros(J), -0.8 + 1.6 * torch.arange(J) / (J - 1), torch.zeros(J)], -1)
        joints = joints + 0.01 * torch.randn(J, 3, generator=g)
    else:
        parents = torch.full((J,), -1, dtype=torch.long)
        joints = torch.zeros(J, 3)
        for i in range(1, J):
            parents[i] = int(torch.randint(0, i, (1,), generator=g))
            joints[i] = joints[parents[i]] + 0.25 * torch.randn(3, generator=g)
        joints = joints / joints.norm(dim=1).max()
    return joints, parents


def make_warp(joints, parents, K):
    with S.quiet():
        sw = SkeletonWarp(is_blender=True, joints=joints, parent_indices=parents, K=K, is_scene_static=True,
                          use_skinning_weight_mlp=False, use_template_offsets=False, hyper_dim=8)
    return sw


def fixture_deform(name, seed, J, N, K, chain=False, mask_random=False, degenerate=False):
    g = torch.Generator().manual_seed(seed)
    joints, parents = random_tree(g, J, chain)
    if degenerate:
        joints[J - 1] = joints[parents[J - 1]]  # zero-length bone
    sw = make_warp(joints, parents, K)
    rho = math.log(0.15) + 0.3 * torch.randn(J, generator=g)
    sw._node_radius.data = rho.clone()
    bone = torch.randint(1, J, (N,), generator=g)
    t = torch.rand(N, 1, generator=g) * 1.4 - 0.2  # some beyond segment ends
    a, b = joints[parents[bone]], joints[bone]
    x = a + t * (b - a) + 0.06 * torch.randn(N, 3, generator=g)
    x[0] = joints[1]  # a point exactly on a joint
    q = torch.tensor([1.0, 0, 0, 0]) + 0.3 * torch.randn(J, 4, generator=g)  # NOT normalised
    gt = 0.02 * torch.randn(3, generator=g)
    mask = torch.sigmoid(torch.randn(N, 1, generator=g)) if mask_random else torch.ones(N, 1)
    q.requires_grad_(True)
    gt.requires_grad_(True)
    mask.requires_grad_(True)
    out = sw.deform_by_pose(x, {"local_rotation": q, "global_trans": gt}, mask)
    # FK pieces (G1)
    R = quaternion_to_matrix(q)
    posed, G = sw.chain_product_transform(R, sw.nodes[:, :3])
    node_rot = matrix_to_quaternion(G[:, :3, :3].detach())
    # distances (G2)
    w, d2, idx = sw.cal_nn_weight_skeleton(x=x, nodes=sw.nodes)
    # backward (G4) with fixed cotangents
    g_xyz = torch.randn(N, 3, generator=g)
    g_rot = torch.randn(N, 4, generator=g)
    g_nodes = torch.randn(J, 3, generator=g)
    loss = (out["d_xyz"] * g_xyz).sum() + (out["d_rotation"] * g_rot).sum() + (out["d_nodes"] * g_nodes).sum()
    loss.backward()
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        joints=np_(joints), parents=np_(parents), node_radius_log=np_(rho), x=np_(x), local_rot=np_(q),
        global_trans=np_(gt), motion_mask=np_(mask), K=np.int64(K),
        R=np_(R), posed=np_(posed), transforms=np_(G), node_rot=np_(node_rot),
        d2=np_(d2), nn_idx=np_(idx), nn_weight=np_(w),
        d_xyz=np_(out["d_xyz"]), d_rotation=np_(out["d_rotation"]), d_scaling=np_(out["d_scaling"]),
        d_nodes=np_(out["d_nodes"]),
        g_xyz=np_(g_xyz), g_rot=np_(g_rot), g_nodes=np_(g_nodes),
        grad_local_rot=np_(q.grad), grad_global_trans=np_(gt.grad), grad_node_radius=np_(sw._node_radius.grad),
        grad_motion_mask=np_(mask.grad),
    )
    print("wrote", name, "J", J, "N", N, "K", K)


def fixture_posemlp(name, seed, J):
    torch.manual_seed(seed)
    net = PoseMLP(1, J * 4, depth=8, hidden_dimensions=32, multires=8)  # small width fits a fixture
    t = torch.tensor([0.37])
    out = net(t)
    sd = {k.replace(".", "__"): np_(v) for k, v in net.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), t=np_(t), rotation=np_(out["rotation"]),
                        translation=np_(out["translation"]), J=np.int64(J), **sd)
    print("wrote", name)


def fixture_glue(name, seed, N, isotropic, from_K):
    """G7: render() glue — what render() hands to the rasterizer, and camera matrices."""
    g = torch.Generator().manual_seed(seed)
    gm = GaussianModel(3, fea_dim=8, with_motion_mask=False, use_isotropic_gs=isotropic)
    P = torch.nn.Parameter
    gm._xyz = P(torch.randn(N, 3, generator=g))
    gm._features_dc = P(torch.randn(N, 1, 3, generator=g))
    gm._features_rest = P(0.1 * torch.randn(N, 15, 3, generator=g))
    gm._scaling = P(math.log(0.05) + 0.35 * torch.randn(N, 1 if isotropic else 3, generator=g))
    gm._rotation = P(torch.randn(N, 4, generator=g))
    gm._opacity = P(1.5 * torch.randn(N, 1, generator=g))
    gm.active_sh_degree = 3
    d_xyz = 0.05 * torch.randn(N, 3, generator=g)
    d_rot = torch.tensor([1.0, 0, 0, 0]) + 0.1 * torch.randn(N, 4, generator=g)
    d_scaling = torch.zeros(N, 3)
    # camera: look-at origin from radius 4, elevation 20 deg, azimuth 45 deg
    az, el, rad = math.radians(45.0), math.radians(20.0), 4.0
    eye = np.array([rad * math.cos(el) * math.sin(az), -rad * math.sin(el), -rad * math.cos(el) * math.cos(az)])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(np.array([0.0, -1.0, 0.0]), fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    Rc2w = np.stack([right, up, fwd], axis=1)  # columns = camera axes in world
    R = Rc2w  # reference stores R = c2w rotation (transposed inside getWorld2View2)
    T = -Rc2w.T @ eye
    H, W = 40, 56
    fov = 0.6911112
    K = None
    if from_K:
        fx = W / (2 * math.tan(fov / 2))
        K = np.array([[fx, 0, W / 2 + 13.0 * W / 1024], [0, fx, H / 2 - 7.0 * H / 1024], [0, 0, 1]], dtype=np.float64)
    cam = Camera(0, R, T, fov, fov * 0.8, torch.zeros(3, H, W), None, "c", 0, data_device="cpu", fid=0.37, K=K)

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False
        debug = False
    bg = torch.zeros(3)
    render(cam, gm, Pipe, bg, d_xyz, d_rot, d_scaling)
    kw, st = S.CAPTURE["kwargs"], S.CAPTURE["settings"]
    # G8: SH colour via the convert_SHs_python branch arithmetic (gaussian_renderer/__init__.py:107-112)
    shs = kw["shs"]
    means3D = kw["means3D"]
    shs_view = shs.transpose(1, 2).view(-1, 3, 16)
    dir_pp = means3D - cam.camera_center.repeat(N, 1)
    dirn = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    cols = {}
    for deg in range(4):
        cols["rgb_deg%d" % deg] = np_(torch.clamp_min(eval_sh(deg, shs_view, dirn) + 0.5, 0.0))
    # G9: Sigma3D via build_scaling_rotation (utils/general_utils.py:137-170) on the
    # rasterizer's own inputs (unit quaternion, activated scales)
    L = build_scaling_rotation(1.0 * kw["scales"], kw["rotations"])
    cov6 = strip_symmetric(L @ L.transpose(1, 2))
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        xyz=np_(gm._xyz), features_dc=np_(gm._features_dc), features_rest=np_(gm._features_rest),
        scaling=np_(gm._scaling), rotation=np_(gm._rotation), opacity=np_(gm._opacity),
        d_xyz=np_(d_xyz), d_rotation=np_(d_rot), d_scaling=np_(d_scaling), isotropic=np.bool_(isotropic),
        cam_R=R, cam_T=T, fovx=np.float64(fov), fovy=np.float64(fov * 0.8), H=np.int64(H), W=np.int64(W),
        K=(K if K is not None else np.zeros((0,), np.float64)),
        means3D=np_(kw["means3D"]), opacities=np_(kw["opacities"]), scales=np_(kw["scales"]),
        rotations=np_(kw["rotations"]), shs=np_(kw["shs"]), means2D=np_(kw["means2D"]),
        viewmatrix=np_(st.viewmatrix), projmatrix=np_(st.projmatrix), campos=np_(st.campos),
        tanfovx=np.float64(st.tanfovx), tanfovy=np.float64(st.tanfovy), sh_degree=np.int64(st.sh_degree),
        cov6=np_(cov6), **cols,
    )
    print("wrote", name)


def fixture_optim(name, seed, N):
    """§8-f rank 1: the Gaussian optimizer exactly as the reference builds and steps it
    (GaussianModel.training_setup scene/gaussian_model.py:197-221 -> torch.optim.Adam(l, lr=0.0, eps=1e-15);
    optimizer.step / update_learning_rate / zero_grad train_rig.py:517-533) and add_densification_stats (:516-518)."""
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(seed)
    gm = GaussianModel(3, fea_dim=0, with_motion_mask=False)
    P = torch.nn.Parameter
    gm._xyz = P(torch.randn(N, 3, generator=g))
    gm._features_dc = P(torch.randn(N, 1, 3, generator=g))
    gm._features_rest = P(0.1 * torch.randn(N, 15, 3, generator=g))
    gm._scaling = P(math.log(0.05) + 0.35 * torch.randn(N, 3, generator=g))
    gm._rotation = P(torch.randn(N, 4, generator=g))
    gm._opacity = P(1.5 * torch.randn(N, 1, generator=g))
    args = SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                           position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05,
                           scaling_lr=0.001, rotation_lr=0.001, skeleton_gs_position_lr=0.00001)
    with S.quiet():
        gm.training_setup(args)
    names = [grp["name"] for grp in gm.optimizer.param_groups]
    out = {"names": np.array(names), "N": N, "betas": np.array(gm.optimizer.param_groups[0]["betas"]),
           "eps": gm.optimizer.param_groups[0]["eps"]}
    for grp in gm.optimizer.param_groups:
        out["p0_" + grp["name"]] = np_(grp["params"][0]).copy()
    steps = 4
    for it in range(steps):
        for grp in gm.optimizer.param_groups:
            p = grp["params"][0]
            # sparse-ish gradients with a wide dynamic range, exact zeros for "invisible" Gaussians
            gr = torch.randn(p.shape, generator=g) * torch.exp(3.0 * torch.randn(p.shape[0], *([1] * (p.dim() - 1)), generator=g))
            gr = gr * (torch.rand(p.shape[0], *([1] * (p.dim() - 1)), generator=g) > 0.3)
            p.grad = gr.clone()
            out["g%d_%s" % (it, grp["name"])] = np_(gr)
        out["lr%d" % it] = np.array([grp["lr"] for grp in gm.optimizer.param_groups], dtype=np.float64)
        gm.optimizer.step()
        gm.update_learning_rate(1000 * (it + 1))
        gm.optimizer.zero_grad(set_to_none=True)
        for grp in gm.optimizer.param_groups:
            st = gm.optimizer.state[grp["params"][0]]
            out["p%d_%s" % (it + 1, grp["name"])] = np_(grp["params"][0]).copy()
            out["m%d_%s" % (it + 1, grp["name"])] = np_(st["exp_avg"]).copy()
            out["v%d_%s" % (it + 1, grp["name"])] = np_(st["exp_avg_sq"]).copy()
    out["steps"] = steps
    # densification statistics, two frames
    for it in range(2):
        vt = SimpleNamespace(grad=torch.randn(N, 3, generator=g) * 1e-3)
        filt = torch.rand(N, generator=g) > 0.4
        gm.add_densification_stats(vt, filt)
        out["ds_grad%d" % it] = np_(vt.grad)
        out["ds_filter%d" % it] = np_(filt)
        out["ds_accum%d" % it] = np_(gm.xyz_gradient_accum).copy()
        out["ds_denom%d" % it] = np_(gm.denom).copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def fixture_loss(name, seed, C, H, W):
    """§8-f rank 2: the image loss exactly as the trainer evaluates it (train_rig.py:508-509) with the reference's own
    l1_loss / ssim (utils/loss_utils.py) and autograd's gradient w.r.t. the rendered image."""
    from utils.loss_utils import l1_loss, ssim
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(C, H, W, generator=g)
    gt[:, : H // 3] = 0.0  # a flat black region (background): mu = sigma = 0 there
    img = (gt + 0.15 * torch.randn(C, H, W, generator=g)).clamp(0, 1)
    img[:, -3:, :] = gt[:, -3:, :]  # exact agreement on a strip: sign(0) = 0 in the L1 gradient
    out = {"image": np_(img), "gt": np_(gt)}
    x = img.clone().requires_grad_(True)
    Ll1, s = l1_loss(x, gt), ssim(x, gt)
    lam = 0.2
    loss = (1.0 - lam) * Ll1 + lam * (1.0 - s)
    loss.backward()
    out.update(l1=float(Ll1.detach()), ssim=float(s.detach()), loss=float(loss.detach()), lambda_dssim=lam, grad_loss=np_(x.grad))
    for key, fn in (("grad_l1", l1_loss), ("grad_ssim", ssim)):
        x = img.clone().requires_grad_(True)
        fn(x, gt).backward()
        out[key] = np_(x.grad)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def chamfer_l1_published(x, y, norm=1):
    """Stand-in for pytorch3d.loss.chamfer_distance (pytorch3d is not vendored in the reference tree and not installed):
    the PUBLISHED algorithm for the call the trainer makes (train_rig.py:313: norm=1, defaults otherwise) — for every
    point the L1 distance to its nearest neighbour in the other set, averaged per set, the two directions added; batch of
    one.  The chamfer part of the fixture is therefore a restatement ("parity unpinned" for that factor); the bone
    sampling, the projection and the composition around it are the reference's own code."""
    assert norm == 1 and x.shape[0] == 1 and y.shape[0] == 1
    d = (x[0][:, None, :] - y[0][None, :, :]).abs().sum(-1)
    return d.min(1).values.mean() + d.min(0).values.mean(), None


def fixture_skeleton_projection(name, seed, J, M, from_K, chain=False):
    """§8-f rank 2 (second half): TrainRig.cal_skeleton_loss (train_rig.py:309-314) = sampling_skeleton_points (:264-276)
    -> project_nodes_to_2d_elements (utils/other_utils.py:101-127) -> chamfer distance to the camera's thinned silhouette
    pixels, with autograd's gradient w.r.t. the posed joints."""
    import types
    class _Inert:  # absorbs the import-time side effects of modules the trainer pulls in (LPIPS nets, GUI)
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Inert()

        def __getattr__(self, n):
            return _Inert()

    lp = types.ModuleType("lpips")
    lp.LPIPS = _Inert
    sys.modules.setdefault("lpips", lp)
    for missing in ("piq", "pytorch_msssim", "dearpygui", "dearpygui.dearpygui"):
        if missing not in sys.modules:
            m = types.ModuleType(missing)
            m.__path__ = []
            m.__getattr__ = lambda n: _Inert
            sys.modules[missing] = m
    with S.quiet():
        import train_rig
        from utils.other_utils import project_nodes_to_2d_elements
    train_rig.chamfer_distance = chamfer_l1_published
    g = torch.Generator().manual_seed(seed)
    joints, parents = random_tree(g, J, chain=chain)
    nodes = (joints + 0.02 * torch.randn(J, 3, generator=g)).requires_grad_(True)
    az, el, rad = math.radians(30.0), math.radians(15.0), 3.5
    eye = np.array([rad * math.cos(el) * math.sin(az), -rad * math.sin(el), -rad * math.cos(el) * math.cos(az)])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(np.array([0.0, -1.0, 0.0]), fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    Rc2w = np.stack([right, up, fwd], axis=1)
    T = -Rc2w.T @ eye
    H, W = 120, 160
    fovx = 0.6911112
    K = None
    if from_K:
        fx = W / (2 * math.tan(fovx / 2))
        K = np.array([[fx, 0, W / 2 + 5.5], [0, fx, H / 2 - 3.25], [0, 0, 1]], dtype=np.float64)
    cam = Camera(0, Rc2w, T, fovx, fovx * 0.8, torch.zeros(3, H, W), None, "c", 0, data_device="cpu", fid=0.5, K=K)
    # thinned silhouette pixels (row, col): near the projected skeleton, jittered, some far outliers, one exact repeat
    with torch.no_grad():
        proj0 = project_nodes_to_2d_elements(cam, nodes.detach())
    pick = torch.randint(0, J, (M,), generator=g)
    thinned = (proj0[pick] + 6.0 * torch.randn(M, 2, generator=g)).round()
    thinned[: M // 10] = torch.stack([torch.randint(0, H, (M // 10,), generator=g),
                                      torch.randint(0, W, (M // 10,), generator=g)], -1).float()
    thinned[-1] = thinned[0]
    cam.thinned = thinned
    fake = types.SimpleNamespace(
        sampling_skeleton_points=lambda j, p: train_rig.TrainRig.sampling_skeleton_points(None, j, p),
        skeleton=types.SimpleNamespace(deform=types.SimpleNamespace(parents=parents)))
    pts = train_rig.TrainRig.sampling_skeleton_points(None, nodes, parents)
    proj = project_nodes_to_2d_elements(cam, pts)
    loss = train_rig.TrainRig.cal_skeleton_loss(fake, nodes, cam)
    loss.backward()
    out = dict(d_nodes=np_(nodes), parents=parents.numpy().astype(np.int32), world_view_transform=np_(cam.world_view_transform),
               FoVx=cam.FoVx, FoVy=cam.FoVy, image_height=H, image_width=W, K=(np.zeros((0, 0)) if K is None else np.asarray(K)),
               thinned=np_(thinned), sampling_points=np_(pts), projected=np_(proj), loss=float(loss.detach()),
               grad_nodes=np_(nodes.grad), steps=pts.shape[0] // (J - 1))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "steps", out["steps"], "points", pts.shape[0], "loss", out["loss"])


def knn_points_published(p1, p2, lengths1=None, lengths2=None, K=1, **kw):
    """Stand-in for pytorch3d.ops.knn_points (not vendored, not installed): the published contract — for every point of p1 the
    K nearest points of p2 by SQUARED Euclidean distance, ascending; returns (dists, idx, None).  "Parity unpinned" for this
    factor; everything around it in the fixture is the reference's own code."""
    d = ((p1[0][:, None, :] - p2[0][None, :, :]) ** 2).sum(-1)
    dist, idx = d.topk(K, dim=1, largest=False, sorted=True)
    return dist[None], idx[None], None


def fixture_control_nodes(name, seed, N, M, K, hyper, local_frame, d_rot_as_res, with_node_weight, mask_random=False):
    """§8-f rank 4 (second half): ControlNodeWarp.forward (utils/time_utils.py:1133-1236) = cal_nn_weight (:934-964) + the
    per-Gaussian blend of the node deformations, with autograd's gradients w.r.t. every differentiable input.  The node
    network is bypassed through the reference's own ``animation_d_values`` hook (:1141-1144): the fixture supplies the node
    attributes a network would predict."""
    import pytorch3d.ops as p3o
    import pytorch3d as p3
    p3o.knn_points = knn_points_published
    p3.ops = p3o
    with S.quiet():
        from utils.time_utils import ControlNodeWarp
        cn = ControlNodeWarp(is_blender=True, node_num=M, K=K, with_node_weight=with_node_weight, local_frame=local_frame,
                             d_rot_as_res=d_rot_as_res, hyper_dim=hyper, is_scene_static=True)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 3, generator=g) * 0.5
    pick = torch.randint(0, N, (M,), generator=g)
    nodes = torch.cat([x[pick] + 0.05 * torch.randn(M, 3, generator=g), 1e-2 + 0.02 * torch.randn(M, hyper, generator=g)], -1)
    cn.nodes = torch.nn.Parameter(nodes)
    cn._node_radius = torch.nn.Parameter(math.log(0.15) + 0.3 * torch.randn(M, generator=g))
    if with_node_weight:
        cn._node_weight = torch.nn.Parameter(0.5 * torch.randn(M, 1, generator=g))
    feature = (0.02 * torch.randn(N, hyper + 1, generator=g)).requires_grad_(True) if hyper > 0 else None
    mask = torch.rand(N, 1, generator=g) if mask_random else torch.ones(N, 1)
    mask = mask.requires_grad_(True)
    attrs = {"d_xyz": 0.1 * torch.randn(M, 3, generator=g), "d_rotation": 0.2 * torch.randn(M, 4, generator=g),
             "d_scaling": 0.05 * torch.randn(M, 3, generator=g), "local_rotation": 0.3 * torch.randn(M, 4, generator=g)}
    attrs = {k: v.requires_grad_(True) for k, v in attrs.items()}
    cn.train()
    out = cn(x, torch.tensor(0.3), feature, mask, animation_d_values=attrs)
    go = {k: torch.randn(out[k].shape, generator=g) for k in ("d_xyz", "d_rotation", "d_scaling", "d_nodes")}
    sum((out[k] * go[k]).sum() for k in go).backward()
    nn_weight, nn_dist, nn_idx = cn.cal_nn_weight(x=x, feature=feature)
    z = dict(x=np_(x), nodes=np_(nodes), _node_radius=np_(cn._node_radius), feature=(np_(feature) if feature is not None else np.zeros((0, 0), np.float32)),
             motion_mask=np_(mask), K=K, hyper_dim=hyper, local_frame=local_frame, d_rot_as_res=d_rot_as_res,
             with_node_weight=with_node_weight, nn_idx=nn_idx.numpy().astype(np.int32), nn_weight=np_(nn_weight), nn_dist=np_(nn_dist))
    if with_node_weight:
        z["_node_weight"] = np_(cn._node_weight)
        z["grad__node_weight"] = np_(cn._node_weight.grad)
    for k, v in attrs.items():
        z["attr_" + k] = np_(v)
        z["grad_attr_" + k] = np_(v.grad) if v.grad is not None else np.zeros(v.shape, np.float32)
    for k in go:
        z["out_" + k] = np_(out[k])
        z["gout_" + k] = np_(go[k])
    z["grad_nodes"] = np_(cn.nodes.grad)
    z["grad__node_radius"] = np_(cn._node_radius.grad)
    z["grad_motion_mask"] = np_(mask.grad)
    if feature is not None:
        z["grad_feature"] = np_(feature.grad)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **z)
    print("wrote", name, {k: float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad_")})


def fixture_control_nodes_blends(name, seed, N, M, skinning, d_rot_as_res):
    """ControlNodeWarp.forward in the modes the shipped stage-1 recipe leaves off: ``skinning=True`` (softmax of a per-Gaussian
    (N, M) feature instead of KNN weights, utils/time_utils.py:934-938) and ``pred_opacity`` / ``pred_color`` (:1214-1225), with
    autograd's gradients."""
    import pytorch3d.ops as p3o
    import pytorch3d as p3
    p3o.knn_points = knn_points_published
    p3.ops = p3o
    with S.quiet():
        from utils.time_utils import ControlNodeWarp
        cn = ControlNodeWarp(is_blender=True, node_num=M, K=3, with_node_weight=True, local_frame=False, d_rot_as_res=d_rot_as_res,
                             hyper_dim=0 if skinning else 2, is_scene_static=True, skinning=skinning, pred_opacity=True,
                             pred_color=True)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 3, generator=g) * 0.5
    hyper = 0 if skinning else 2
    nodes = torch.cat([x[torch.randint(0, N, (M,), generator=g)] + 0.05 * torch.randn(M, 3, generator=g),
                       1e-2 + 0.02 * torch.randn(M, hyper, generator=g)], -1)
    cn.nodes = torch.nn.Parameter(nodes)
    if not skinning:
        cn._node_radius = torch.nn.Parameter(math.log(0.15) + 0.3 * torch.randn(M, generator=g))
        cn._node_weight = torch.nn.Parameter(0.5 * torch.randn(M, 1, generator=g))
    feature = (torch.randn(N, M, generator=g) if skinning else 0.02 * torch.randn(N, hyper + 1, generator=g)).requires_grad_(True)
    mask = torch.rand(N, 1, generator=g).requires_grad_(True)
    attrs = {"d_xyz": 0.1 * torch.randn(M, 3, generator=g), "d_rotation": 0.2 * torch.randn(M, 4, generator=g),
             "d_scaling": 0.05 * torch.randn(M, 3, generator=g), "local_rotation": 0.3 * torch.randn(M, 4, generator=g),
             "d_opacity": 0.3 * torch.randn(M, 1, generator=g), "d_color": 0.2 * torch.randn(M, 3, generator=g)}
    attrs = {k: v.requires_grad_(True) for k, v in attrs.items()}
    cn.train()
    out = cn(x, torch.tensor(0.3), feature, mask, animation_d_values=attrs)
    keys = ("d_xyz", "d_rotation", "d_scaling", "d_opacity", "d_color")
    go = {k: torch.randn(out[k].shape, generator=g) for k in keys}
    sum((out[k] * go[k]).sum() for k in go).backward()
    z = dict(x=np_(x), nodes=np_(nodes), feature=np_(feature), motion_mask=np_(mask), skinning=skinning, d_rot_as_res=d_rot_as_res,
             grad_feature=np_(feature.grad), grad_motion_mask=np_(mask.grad))
    if not skinning:
        z.update(_node_radius=np_(cn._node_radius), _node_weight=np_(cn._node_weight), grad__node_radius=np_(cn._node_radius.grad),
                 grad__node_weight=np_(cn._node_weight.grad), grad_nodes=np_(cn.nodes.grad))
    for k, v in attrs.items():
        z["attr_" + k] = np_(v)
        z["grad_attr_" + k] = np_(v.grad) if v.grad is not None else np.zeros(v.shape, np.float32)
    for k in keys:
        z["out_" + k] = np_(out[k])
        z["gout_" + k] = np_(go[k])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **z)
    print("wrote", name, {k: float(np.abs(v).max()) for k, v in z.items() if k.startswith("grad_")})


class WavingNodes(torch.nn.Module):
    """A closed-form node network for the editing fixtures (the reference asks its network for the nodes' trajectory at four
    times, time_utils.py:1004-1011): d_xyz = 0.08 sin(2 pi t + 3 x), the other attributes zero.  tests/test_gpu_cnode.py builds
    the same module."""

    def forward(self, x, t, **kwargs):
        z3, z4 = torch.zeros_like(x), torch.zeros(x.shape[0], 4, dtype=x.dtype, device=x.device)
        return {"d_xyz": 0.08 * torch.sin(6.283185307179586 * t + 3.0 * x), "d_rotation": z4, "d_scaling": z3,
                "local_rotation": z4.clone(), "hidden": None, "d_opacity": None, "d_color": None}


def fixture_control_nodes_edit(name, seed, N, M, d_rot_as_res):
    """ControlNodeWarp.forward with ``node_trans_bias`` (utils/time_utils.py:1165-1213, the GUI's drag-to-edit path, under
    no_grad): p2dR (:1044-1077, trajectory mode through get_trajectory :1004-1011 and torch.svd), cal_nn_weight_floyd /
    geodesic_distance_floyd (:969-988, 1122-1131), cal_nn_weight on the posed nodes with K = 32.  knn_points is the published-
    contract stand-in (see above)."""
    import pytorch3d.ops as p3o
    import pytorch3d as p3
    p3o.knn_points = knn_points_published
    p3.ops = p3o
    with S.quiet():
        from utils.time_utils import ControlNodeWarp
        cn = ControlNodeWarp(is_blender=True, node_num=M, K=3, with_node_weight=True, local_frame=False, d_rot_as_res=d_rot_as_res,
                             hyper_dim=2, is_scene_static=True)
    cn.network = WavingNodes()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 3, generator=g) * 0.5
    nodes = torch.cat([x[torch.randint(0, N, (M,), generator=g)] + 0.05 * torch.randn(M, 3, generator=g),
                       1e-2 + 0.02 * torch.randn(M, 2, generator=g)], -1)
    cn.nodes = torch.nn.Parameter(nodes)
    cn._node_radius = torch.nn.Parameter(math.log(0.15) + 0.3 * torch.randn(M, generator=g))
    cn._node_weight = torch.nn.Parameter(0.5 * torch.randn(M, 1, generator=g))
    feature = 0.02 * torch.randn(N, 3, generator=g)
    mask = torch.rand(N, 1, generator=g)
    # a drag: a handful of nodes pulled along one direction, their neighbours less
    pull = torch.zeros(M, 3)
    centre = nodes[0, :3]
    pull[:] = torch.tensor([0.25, -0.1, 0.15]) * torch.exp(-((nodes[:, :3] - centre) ** 2).sum(-1, keepdim=True) / 0.08)
    t = torch.tensor(0.3)
    with torch.no_grad():
        out = cn(x, t, feature, mask, node_trans_bias=pull)
    z = dict(x=np_(x), nodes=np_(nodes), _node_radius=np_(cn._node_radius), _node_weight=np_(cn._node_weight), feature=np_(feature),
             motion_mask=np_(mask), node_trans_bias=np_(pull), t=float(t), d_rot_as_res=d_rot_as_res,
             out_d_xyz=np_(out["d_xyz"]), out_d_rotation=np_(out["d_rotation"]), out_d_scaling=np_(out["d_scaling"]))
    if "d_rotation_bias" in out:
        z["out_d_rotation_bias"] = np_(out["d_rotation_bias"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **z)
    print("wrote", name, {k: float(np.abs(v).max()) for k, v in z.items() if k.startswith("out_")})


def seeded_heads(J, WeightCls, DeformCls, seed):
    """The two per-Gaussian MLP heads with reproducible weights (the fixture stores checksums, not 4 MB of weights):
    constructed standalone, in this order, right after torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    wm = WeightCls(input_ch=3, output_ch=J - 1)
    dn = DeformCls(xyz_input_ch=3, time_input_ch=J * 4, t_multires=-1)
    with torch.no_grad():
        dn.gaussian_warp.weight.mul_(2000.0)  # the reference initialises this head at std 1e-5: make the offsets visible
        dn.gaussian_warp.bias.add_(0.01)
    return wm, dn


def fixture_deform_heads(name, seed, J, N):
    """§8-f rank 3: deform_by_pose with BOTH per-Gaussian MLP heads on (use_skinning_weight_mlp, use_template_offsets:
    skeleton_warp.py:24-32, 56-69, 152-158), K = -1, values and gradients incl. into the heads' parameters."""
    from skeleton_utils.network_utils import DeformMLP, WeightMLP
    g = torch.Generator().manual_seed(seed)
    joints, parents = random_tree(g, J)
    with S.quiet():
        sw = SkeletonWarp(is_blender=True, joints=joints, parent_indices=parents, K=-1, is_scene_static=True,
                          use_skinning_weight_mlp=True, use_template_offsets=True, hyper_dim=8)
    sw.skinning_weight_mlp, sw.detail_net = seeded_heads(J, WeightMLP, DeformMLP, seed + 1000)
    sw._node_radius.data = torch.log(0.15 + 0.2 * torch.rand(J, generator=g))
    x = (joints[torch.randint(0, J, (N,), generator=g)] + 0.12 * torch.randn(N, 3, generator=g))
    q = (torch.tensor([1.0, 0, 0, 0]) + 0.35 * torch.randn(J, 4, generator=g)).requires_grad_(True)
    gt = (0.1 * torch.randn(3, generator=g)).requires_grad_(True)
    mask = torch.rand(N, 1, generator=g)
    out = sw.deform_by_pose(x, {"local_rotation": q, "global_trans": gt}, mask)
    c_xyz, c_rot = torch.randn(N, 3, generator=g), torch.randn(N, 4, generator=g)
    ((out["d_xyz"] * c_xyz).sum() + (out["d_rotation"] * c_rot).sum()).backward()
    heads = {"wm": sw.skinning_weight_mlp, "dn": sw.detail_net}
    picks = {"wm": ["linear.0.weight", "linear.5.bias", "weight_predict.weight"], "dn": ["linear.0.weight", "linear.5.bias", "gaussian_warp.weight"]}
    res = dict(joints=np_(joints), parents=np_(parents), x=np_(x), node_radius=np_(sw._node_radius), local_rot=np_(q), global_trans=np_(gt),
               mask=np_(mask), head_seed=seed + 1000, c_xyz=np_(c_xyz), c_rot=np_(c_rot), d_xyz=np_(out["d_xyz"]),
               d_rotation=np_(out["d_rotation"]), nn_weight=np_(out["nn_weight"]), template_offsets=np_(sw.template_offsets),
               skinning_weight_offsets=np_(sw.skinning_weight_offsets), g_local_rot=np_(q.grad), g_global_trans=np_(gt.grad),
               g_node_radius=np_(sw._node_radius.grad))
    for hk, mod in heads.items():
        sd = dict(mod.named_parameters())
        res["chk_" + hk] = np.array([float(p.detach().double().abs().sum()) for p in sd.values()])
        for pn in picks[hk]:
            res["g_%s_%s" % (hk, pn)] = np_(sd[pn].grad)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
    print("wrote", name)


def fixture_densify(name, seed, N, isotropic, fea_dim):
    """the reference's own densify_and_prune (scene/gaussian_model.py:500-514 -> densify_and_clone :475,
    densify_and_split :440, prune_points :373 with the optimizer surgery :338-417) and reset_opacity (:275) on a model that
    has taken one Adam step (so the moments are non-trivial) and has densification statistics; plus prune_points /
    densify_and_clone / densify_and_split called on their own.  torch.normal is wrapped to record the unit normals behind the
    split children (the draw is samples = std * z)."""
    from types import SimpleNamespace
    args = SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                           position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05,
                           scaling_lr=0.001, rotation_lr=0.001, skeleton_gs_position_lr=0.00001)
    names = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"] + (["feature"] if fea_dim else [])

    def model():
        g = torch.Generator().manual_seed(seed)
        gm = GaussianModel(3, fea_dim=fea_dim, with_motion_mask=False, use_isotropic_gs=isotropic)
        P = torch.nn.Parameter
        gm._xyz = P(torch.randn(N, 3, generator=g))
        gm._features_dc = P(torch.randn(N, 1, 3, generator=g))
        gm._features_rest = P(0.1 * torch.randn(N, 15, 3, generator=g))
        gm._scaling = P(math.log(0.04 if isotropic else 0.015) + 0.9 * torch.randn(N, 1 if isotropic else 3, generator=g))
        gm._rotation = P(torch.randn(N, 4, generator=g))
        gm._opacity = P(2.5 * torch.randn(N, 1, generator=g))
        if fea_dim:
            gm.feature = P(torch.randn(N, fea_dim, generator=g))
        gm.max_radii2D = torch.rand(N, generator=g) * 40
        with S.quiet():
            gm.training_setup(args)
        for grp in gm.optimizer.param_groups:
            grp["params"][0].grad = torch.randn(grp["params"][0].shape, generator=g)
        gm.optimizer.step()
        gm.optimizer.zero_grad(set_to_none=True)
        gm.xyz_gradient_accum = (torch.rand(N, 1, generator=g) ** 3) * 6e-4 * torch.randint(0, 4, (N, 1), generator=g)
        gm.denom = torch.randint(0, 4, (N, 1), generator=g).float()   # zeros among them: 0 / 0 = NaN -> 0 (:502)
        return gm, g

    def snap(gm, tag, out):
        grp = {g_["name"]: g_ for g_ in gm.optimizer.param_groups}
        for k in names:
            p = grp[k]["params"][0]
            st = gm.optimizer.state[p]
            out["%s_%s" % (tag, k)] = np_(p).copy()
            out["%s_m_%s" % (tag, k)] = np_(st["exp_avg"]).copy()
            out["%s_v_%s" % (tag, k)] = np_(st["exp_avg_sq"]).copy()
            out["%s_step_%s" % (tag, k)] = np.array(float(st["step"]))
        out[tag + "_accum"], out[tag + "_denom"], out[tag + "_radii2D"] = np_(gm.xyz_gradient_accum).copy(), np_(gm.denom).copy(), np_(gm.max_radii2D).copy()
    out = {"isotropic": np.array(isotropic), "fea_dim": np.array(fea_dim), "percent_dense": np.array(args.percent_dense)}
    rec = []
    orig_normal = torch.normal

    def normal(mean=None, std=None, **kw):
        z = torch.randn(std.shape, generator=torch.Generator().manual_seed(seed + 7 + len(rec)))
        rec.append(z)
        return mean + std * z
    torch.normal = normal
    try:
        # (1) densify_and_prune as the trainer calls it (train_rig.py:361): with and without the screen-size argument
        for tag, extent, screen in (("dp", 2.0, 20), ("dq", 2.0, None)):
            gm, g = model()
            snap(gm, tag + "0", out)
            rec.clear()
            with S.quiet():
                gm.densify_and_prune(0.0002, 0.005, extent, screen)
            snap(gm, tag + "1", out)
            out[tag + "_z"] = np_(rec[0])
            out[tag + "_args"] = np.array([0.0002, 0.005, extent, -1.0 if screen is None else float(screen)])
        # (2) the pieces on their own
        gm, g = model()
        mask = torch.rand(N, generator=g) < 0.3
        gm.prune_points(mask)
        snap(gm, "pr1", out)
        out["pr_mask"] = np_(mask)
        gm, g = model()
        grads = gm.xyz_gradient_accum / gm.denom
        grads[grads.isnan()] = 0.0
        gm.densify_and_clone(grads, 0.0002, 2.0)
        snap(gm, "cl1", out)
        gm, g = model()
        grads = gm.xyz_gradient_accum / gm.denom
        grads[grads.isnan()] = 0.0
        rec.clear()
        gm.densify_and_split(grads, 0.0002, 2.0)
        snap(gm, "sp1", out)
        out["sp_z"] = np_(rec[0])
        # (3) reset_opacity
        gm, g = model()
        gm.reset_opacity()
        snap(gm, "ro1", out)
    finally:
        torch.normal = orig_normal
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: out[k].shape for k in ("dp0_xyz", "dp1_xyz", "dq1_xyz", "pr1_xyz", "cl1_xyz", "sp1_xyz")})


def fixture_dqb(name, seed, N, K, mode, rot_as_q):
    """Dual-quaternion blending exactly as the reference computes it (utils/dual_quaternion.py: QT2DQ :135, DQ2QT :146,
    DQBlending :168, interpolate :182, transformation_blending :190) with autograd's gradients w.r.t. (q, t, weights).
    mode: "shared2d"  q (K, 4), t (K, 3), weights (N, K)            -> DQBlending (per-quaternion normalisation)
          "shared3d"  q (1, K, 4)                                   -> DQBlending (F.normalize acts on the node axis)
          "rows3d"    q (N, K, 4), t (N, K, 3), weights (N, K)      -> DQBlending, every row its own K transforms
          "interp"    q0, q1 (N, 4), weight (N, 1)                  -> interpolate
          "tblend"    transformations (K, 4, 4), weights (N, K)     -> transformation_blending (N, 4, 4)"""
    import utils.dual_quaternion as DQ
    g = torch.Generator().manual_seed(seed)
    R = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    out = {"mode": np.array(mode), "rot_as_q": np.array(rot_as_q)}
    w = torch.rand(N, K, generator=g) ** 3 + 1e-3
    w = (w / w.sum(-1, keepdim=True)).requires_grad_(True)
    if mode == "tblend":
        q = torch.nn.functional.normalize(R(K, 4), dim=-1)
        T = torch.eye(4)[None].repeat(K, 1, 1)
        T[:, :3, :3] = DQ.quaternion_to_matrix(q)
        T[:, :3, 3] = 0.4 * R(K, 3)
        T = T.requires_grad_(True)
        res = DQ.transformation_blending(T, w)
        gout = R(N, 4, 4)
        (res * gout).sum().backward()
        out.update(transformations=np_(T), weights=np_(w), out=np_(res), gout=np_(gout), grad_transformations=np_(T.grad),
                   grad_weights=np_(w.grad))
    elif mode == "interp":
        q0, q1 = (R(N, 4) * (0.5 + torch.rand(N, 1, generator=g))).requires_grad_(True), (R(N, 4) * 1.3).requires_grad_(True)
        t0, t1 = (0.5 * R(N, 3)).requires_grad_(True), (0.5 * R(N, 3)).requires_grad_(True)
        wt = torch.rand(N, 1, generator=g).requires_grad_(True)
        rot, t_ = DQ.interpolate(q0, t0, q1, t1, wt, rot_as_q=rot_as_q)
        g_rot, g_t = R(*rot.shape), R(N, 3)
        ((rot * g_rot).sum() + (t_ * g_t).sum()).backward()
        out.update(q0=np_(q0), t0=np_(t0), q1=np_(q1), t1=np_(t1), weight=np_(wt), out_rot=np_(rot), out_t=np_(t_), g_rot=np_(g_rot),
                   g_t=np_(g_t), grad_q0=np_(q0.grad), grad_t0=np_(t0.grad), grad_q1=np_(q1.grad), grad_t1=np_(t1.grad),
                   grad_weight=np_(wt.grad))
    else:
        shape = {"shared2d": (K,), "shared3d": (1, K), "rows3d": (N, K)}[mode]
        q = (R(*shape, 4) * (0.6 + torch.rand(*shape, 1, generator=g))).requires_grad_(True)   # non-unit on purpose
        t = (0.5 * R(*shape, 3)).requires_grad_(True)
        rot, t_ = DQ.DQBlending(q, t, w, rot_as_q=rot_as_q)
        g_rot, g_t = R(*rot.shape), R(N, 3)
        ((rot * g_rot).sum() + (t_ * g_t).sum()).backward()
        out.update(q=np_(q), t=np_(t), weights=np_(w), out_rot=np_(rot), out_t=np_(t_), g_rot=np_(g_rot), g_t=np_(g_t),
                   grad_q=np_(q.grad), grad_t=np_(t.grad), grad_weights=np_(w.grad))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def fixture_state_dict_layout():
    """§8-f rank 4a: names and shapes of the reference SkeletonWarp's state dict (what skeleton.pth holds), J = 6,
    hyper_dim = 8, with its static and its non-static base network."""
    import json
    joints, parents = torch.rand(6, 3), torch.tensor([-1, 0, 1, 1, 3, 0])
    out = {}
    for static in (True, False):
        with S.quiet():
            sw = SkeletonWarp(is_blender=True, joints=joints, parent_indices=parents, K=-1, is_scene_static=static, hyper_dim=8)
        out["static" if static else "dynamic"] = {k: list(v.shape) for k, v in sw.state_dict().items()}
    json.dump(out, open(os.path.join(HERE, "skeleton_state_dict_layout.json"), "w"), indent=0)


if __name__ == "__main__":
    fixture_deform("deform_chain8_n257", 11, 8, 257, -1, chain=True)
    fixture_deform("deform_tree24_n1024", 12, 24, 1024, -1, mask_random=True)
    fixture_deform("deform_tree32_n512_k3", 13, 32, 512, 3)
    fixture_deform("deform_tree64_n300", 14, 64, 300, -1)
    fixture_deform("deform_tree6_degenerate", 15, 6, 129, -1, degenerate=True, mask_random=True)
    fixture_posemlp("posemlp_w32_j24", 21, 24)
    fixture_glue("glue_aniso_fov", 31, 96, False, False)
    fixture_glue("glue_iso_K", 32, 96, True, True)
    fixture_optim("optim_adam_n67", 41, 67)
    fixture_loss("loss_l1_ssim", 51, 3, 37, 45)
    fixture_deform_heads("heads_tree12_n200", 61, 12, 200)
    fixture_skeleton_projection("skelproj_tree24_m700", 71, 24, 700, False)
    fixture_control_nodes("cnodes_local_res_h8", 81, 400, 64, 3, 8, True, True, True, mask_random=True)
    fixture_control_nodes("cnodes_global_abs_h0", 82, 257, 40, 4, 0, False, False, False)
    fixture_control_nodes("cnodes_default_h8", 83, 300, 128, 3, 8, False, True, True)
    fixture_control_nodes_blends("cnodes_skinning_m48", 84, 220, 48, True, True)
    fixture_control_nodes_blends("cnodes_skinning_abs_m32", 85, 150, 32, True, False)
    fixture_control_nodes_blends("cnodes_knn_pred_opacity_color", 86, 260, 64, False, True)
    fixture_control_nodes_edit("cnodes_edit_res_m96", 87, 500, 96, True)
    fixture_control_nodes_edit("cnodes_edit_abs_m64", 88, 350, 64, False)
    fixture_skeleton_projection("skelproj_chain8_m90_K", 72, 8, 90, True, chain=True)
    fixture_densify("densify_aniso_n96", 101, 96, False, 0)
    fixture_densify("densify_iso_fea9_n80", 102, 80, True, 9)
    fixture_dqb("dqb_shared2d_k23_q", 91, 300, 23, "shared2d", True)
    fixture_dqb("dqb_shared3d_k63_R", 92, 257, 63, "shared3d", False)
    fixture_dqb("dqb_rows3d_k3_q", 93, 400, 3, "rows3d", True)
    fixture_dqb("dqb_rows3d_k8_R", 94, 129, 8, "rows3d", False)
    fixture_dqb("dqb_interp_q", 95, 200, 2, "interp", True)
    fixture_dqb("dqb_tblend_k31", 96, 150, 31, "tblend", True)
    fixture_dqb("dqb_shared2d_k100_R", 97, 130, 100, "shared2d", False)
    fixture_state_dict_layout()
