"""Record what the REFERENCE's own callers do to the skeleton objects (build container only — imports /root/reference):

    python tests/golden/record_api.py      ->  tests/golden/skeleton_api_calls.json

The reference's ``SkeletonModel`` (scene/skeleton_model.py) is built around the reference's ``SkeletonWarp``; both are
wrapped in the recording proxies of tests/api_replay.py; then the reference's OWN code runs against them, unmodified:

  * ``TrainRig.train_step`` (train_rig.py:535-554) and everything under it — ``select_random_cam``, ``deform_gaussians``
    (:386-414, the flag toggles at iterations < / == / > ``optimize_template_offsets_iters``), ``render_and_cal_loss``
    (:416-515), ``report_and_densification`` (:317-365, incl. ``save_weights``), ``optimizer_step`` (:517-533) — on a
    ``TrainRig`` object made with ``__new__`` (its ``__init__`` needs a dataset on disk; the four lines of it that touch the
    skeleton, :84-92, are transcribed below and marked as such);
  * ``render_rig.render_set`` (:111-218) and ``render_rig.generate_random_motion`` (:250-334), with the image writers /
    metric networks of that module replaced by inert stand-ins;
  * ``GUI.test_step`` (interactive_GUI.py:348-667) in its skeleton-only and its deformed-Gaussians mode, and
    ``GUI.animation_initialize``'s first line (:265).

Only data leaves this script: the list of (who, path, op, name, argument / result descriptions).  The rasterizer is the
capturing stand-in of _ref_shim.py (zeros): nothing here depends on pixels.
"""
import json
import math
import os
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_shim as S  # noqa: E402

S.install()
from tests import api_replay as A  # noqa: E402


class _Inert(types.ModuleType):
    """A module that is absent from this image and only needed for its import side (LPIPS nets, the GUI toolkit, video
    writers): any attribute is an inert mock."""

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        m = MagicMock(name=self.__name__ + "." + n)
        setattr(self, n, m)
        return m


for _m in ("lpips", "piq", "pytorch_msssim", "dearpygui", "dearpygui.dearpygui", "torchvision", "torchvision.utils", "tensorboard",
           "torch.utils.tensorboard"):
    if _m not in sys.modules:
        try:
            __import__(_m)
        except Exception:
            sys.modules[_m] = _Inert(_m)

with S.quiet():
    import train_rig
    import render_rig
    import interactive_GUI
    from arguments import ModelParams, OptimizationParams, PipelineParams
    from scene.cameras import Camera
    from scene.gaussian_model import GaussianModel
    from scene.skeleton_model import SkeletonModel
    from make_golden import chamfer_l1_published, random_tree

SEED = 7
REF = "/root/reference/"
_CITE = [None]


def locate():
    """file:line of the nearest frame inside the reference (or the transcription label in force)."""
    f = sys._getframe(1)
    while f is not None:
        fn = f.f_code.co_filename
        if fn.startswith(REF):
            return "%s:%d" % (fn[len(REF):], f.f_lineno)
        f = f.f_back
    return _CITE[0] or "record_api.py"


class cite:
    def __init__(self, label):
        self.label = label

    def __enter__(self):
        _CITE[0] = self.label + " (transcribed)"

    def __exit__(self, *a):
        _CITE[0] = None


def look_at(az_deg, H, W, fid, uid):
    az, el, rad = math.radians(az_deg), math.radians(20.0), 4.0
    eye = np.array([rad * math.cos(el) * math.sin(az), -rad * math.sin(el), -rad * math.cos(el) * math.cos(az)])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(np.array([0.0, -1.0, 0.0]), fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    R = np.stack([right, up, fwd], axis=1)
    T = -R.T @ eye
    g = torch.Generator().manual_seed(100 + uid)
    cam = Camera(uid, R, T, 0.6911112, 0.6911112, torch.rand(3, H, W, generator=g), None, "c%d" % uid, uid, data_device="cpu", fid=fid)
    cam.thinned = torch.stack([torch.randint(0, H, (50,), generator=g), torch.randint(0, W, (50,), generator=g)], -1).float()
    return cam


def params(group_cls):
    from argparse import ArgumentParser
    p = ArgumentParser()
    grp = group_cls(p)
    return grp.extract(p.parse_args([]))


def main():
    log = []
    J, N, H, W = 8, 48, 32, 32
    g = torch.Generator().manual_seed(SEED)
    joints, parents = random_tree(g, J)
    opt = params(OptimizationParams)
    dataset = params(ModelParams)
    pipe = params(PipelineParams)
    dataset.is_blender = True
    dataset.use_template_offsets = True      # the shipped recipe (scripts/run_demo.py:32): both heads on after iteration 15 000
    dataset.use_skinning_weight_mlp = True
    dataset.load2gpu_on_the_fly = False
    opt.gs_densification_iterations = 10 ** 9  # (densification reads the rasterizer's screen-space gradient: not on this path)
    tmp = tempfile.mkdtemp()
    args = types.SimpleNamespace(model_path=tmp, skinning=False)

    # ---- train_rig.py:84 — the constructor call, keyword for keyword
    ctor = dict(K=opt.skeleton_weight_knn, is_blender=dataset.is_blender, skinning=args.skinning, hyper_dim=dataset.hyper_dim,
                joints=joints, parent_indices=parents, pred_opacity=dataset.pred_opacity, pred_color=dataset.pred_color,
                use_hash=dataset.use_hash, hash_time=dataset.hash_time,
                d_rot_as_res=dataset.d_rot_as_res and not dataset.d_rot_as_rotmat, local_frame=dataset.local_frame,
                progressive_brand_time=dataset.progressive_brand_time, with_arap_loss=not opt.no_arap_loss,
                max_d_scale=dataset.max_d_scale, enable_densify_prune=opt.node_enable_densify_prune,
                is_scene_static=dataset.is_scene_static, use_skinning_weight_mlp=dataset.use_skinning_weight_mlp,
                use_template_offsets=dataset.use_template_offsets)
    with S.quiet():
        model = SkeletonModel(**ctor)
    A.seed_module(model.deform, SEED)
    gs_children = {}
    deform_children = {"as_gaussians": ("gs", gs_children), "gs": ("gs", gs_children)}
    model.deform = A.Recorder(model.deform, "deform", log, locate, deform_children)  # what SkeletonModel's own methods touch
    skeleton = A.Recorder(model, "skeleton", log, locate, {"deform": ("deform", deform_children)})

    # ---- the Gaussians (reference class, small cloud around the bones)
    gaussians = GaussianModel(dataset.sh_degree, fea_dim=dataset.hyper_dim, with_motion_mask=dataset.gs_with_motion_mask,
                              use_isotropic_gs=dataset.use_isotropic_gs)
    bone = torch.randint(1, J, (N,), generator=g)
    t = torch.rand(N, 1, generator=g)
    xyz = joints[parents[bone]] + t * (joints[bone] - joints[parents[bone]]) + 0.05 * torch.randn(N, 3, generator=g)
    P = torch.nn.Parameter
    gaussians._xyz = P(xyz.clone())
    gaussians._features_dc = P(torch.randn(N, 1, 3, generator=g))
    gaussians._features_rest = P(0.1 * torch.randn(N, 15, 3, generator=g))
    gaussians._scaling = P(math.log(0.03) + 0.2 * torch.randn(N, 3, generator=g))
    gaussians._rotation = P(torch.randn(N, 4, generator=g))
    gaussians._opacity = P(torch.randn(N, 1, generator=g))
    gaussians.feature = P(-1e-2 * torch.ones(N, gaussians.fea_dim))
    gaussians.max_radii2D = torch.zeros(N)
    gaussians.active_sh_degree = 3
    gaussians.training_setup(opt)
    cams = [look_at(45.0 * k, H, W, fid=k / 4.0, uid=k) for k in range(4)]

    # ---- TrainRig.__init__, the lines that touch the skeleton (train_rig.py:85-92)
    with cite("train_rig.py:85"):
        with S.quiet():
            skeleton.train_setting(opt)
    with cite("train_rig.py:88"):
        rho = A.seeded_values("_node_radius", (J,), SEED)
        skeleton.deform._node_radius  # (self.skeleton.deform._node_radius.data = deform.deform._node_radius[joint_node_indices])
        log.append({"who": _CITE[0], "path": "deform", "op": "set_data", "name": "_node_radius", "value": A.describe(rho)})
        A.unwrap(model.deform)._node_radius.data = rho.clone()

    # ---- a TrainRig without its __init__; the reference's methods run on it as they are
    train_rig.chamfer_distance = chamfer_l1_published
    train_rig.skeleton_training_report = lambda *a, **k: (None, None, None, None, None)
    rig = train_rig.TrainRig.__new__(train_rig.TrainRig)
    rig.dataset, rig.args, rig.opt, rig.pipe = dataset, args, opt, pipe
    rig.testing_iterations, rig.saving_iterations = [], [15001]
    rig.device = "cpu"
    rig.tb_writer = MagicMock()
    rig.gaussians = gaussians
    rig.scene = types.SimpleNamespace(getTrainCameras=lambda: cams, save=lambda it: None, cameras_extent=1.0, loaded_iter=None)
    rig.skeleton = skeleton
    rig.template_idx = 0
    rig.joints = joints
    rig.background = torch.zeros(3)
    rig.iter_start, rig.iter_end = MagicMock(), MagicMock()
    rig.viewpoint_stack = None
    rig.ema_loss_for_log = 0.0
    rig.best_psnr = rig.best_ssim = rig.best_ms_ssim = 0.0
    rig.best_lpips = rig.best_alex_lpips = np.inf
    rig.best_iteration = 0
    rig.progress_bar = MagicMock()
    rig.smooth_term = lambda it: 0.0
    rig.all_nodes_projection_loss = 1.0e5 * torch.ones(len(cams))
    rig.pretrain_deform_info = {"d_xyz": [torch.zeros(N, 3) for _ in cams], "d_joints": [torch.zeros(J, 3) for _ in cams]}
    rig.test_network_connect = lambda: None  # (the socket viewer: gaussian_renderer/network_gui.py)

    def restore():  # the optimizers moved the parameters: back to the seeded values, so that later results compare by value
        A.seed_module(A.unwrap(model.deform), SEED, keep=("control_nodes",))  # (no optimizer owns them: train_rig.py:404 set them)
        A.unwrap(model.deform)._node_radius.data = rho.clone()

    import random
    for it, warm in ((1, True), (opt.skeleton_warm_up + 5, False), (opt.optimize_template_offsets_iters, False),
                     (opt.optimize_template_offsets_iters + 1, False)):
        rig.iteration = it
        random.seed(it)
        torch.manual_seed(it)
        with S.quiet():
            rig.train_step(warm)
        restore()

    # ---- checkpoints (scene/skeleton_model.py:43-72 <- train_rig.py:92,351; render_rig.py:458)
    with cite("train_rig.py:92"):
        skeleton.load_weights(tmp, iteration=-1)
    with cite("render_rig.py:458"):
        skeleton.load_weights(tmp, iteration=15001)
        skeleton.load_weights(tmp, iteration=12)
    with cite("train_rig.py:352"):  # (commented out there; kept callable)
        with S.quiet():
            d_nodes = A.unwrap(model.deform).get_pose_info(torch.tensor([0.25]).expand(J, 1))
        skeleton.save_joints(tmp, 15001, torch.zeros(J, 3), 3)
    restore()

    # ---- render_rig.render_set / generate_random_motion with the writers and the metric networks replaced
    for n in ("write_to_obj", "vis_blending_weight_all", "imageio", "torchvision"):
        setattr(render_rig, n, MagicMock())
    for n in ("psnr", "ssim_func", "lpips", "ms_ssim", "alex_lpips"):
        setattr(render_rig, n, lambda a, b, **k: torch.zeros(1))
    render_rig.get_color_for_skinning_weights = lambda xyz, vn_idx, vn_weight, control_points: torch.zeros(xyz.shape[0], 3)
    render_rig.project_nodes_to_2d_withnodes = lambda view, nodes, d_nodes, parents, img, path: img
    render_rig.tqdm = lambda x, **k: x
    deform_stage1 = types.SimpleNamespace(d_rot_as_res=True)
    for c in cams:
        c.gt_alpha_mask = torch.ones(1, H, W)
    with S.quiet():
        render_rig.render_set(tmp, False, "test", 15001, cams[:2], gaussians, pipe, torch.zeros(3), deform_stage1, skeleton, 0)
        random.seed(3)
        mark = len(log)
        render_rig.generate_random_motion(tmp, False, "test", 15001, cams, gaussians, pipe, torch.zeros(3), deform_stage1, skeleton, 0)
    # (its 60 poses are the same calls 60 times: the first three stay)
    calls = [i for i in range(mark, len(log)) if log[i]["op"] == "call" and log[i]["name"] == "deform_by_pose"]
    cut = calls[3]
    while log[cut - 1]["who"].startswith("render_rig.py:") and int(log[cut - 1]["who"].split(":")[1].split()[0]) <= 303 \
            and int(log[cut - 1]["who"].split(":")[1].split()[0]) >= 298:
        cut -= 1
    del log[cut:]

    # ---- interactive_GUI.GUI.test_step in its two skeleton modes (interactive_GUI.py:348-667)
    gui = interactive_GUI.GUI.__new__(interactive_GUI.GUI)
    gui.skeleton, gui.gaussians, gui.pipe, gui.background = skeleton, gaussians, pipe, torch.zeros(3)
    gui.scene = rig.scene
    gui.args, gui.opt = args, opt
    gui.is_animation, gui.animation_time, gui.video_speed = True, 0.3, 1.0
    gui.render_interpolation_poses, gui.interpolation_poses = False, None
    gui.should_save_screenshot = gui.should_save_skeleton_pose = gui.should_save_reference_skeleton_pose = False
    gui.edited_skeleton_pose = gui.saved_skeleton_pose = gui.motion_animation_d_values = None
    gui.view_animation, gui.animation_trans_bias = False, None
    gui.vis_scale_const, gui.vis_traj_realtime, gui.showing_overlay = None, False, False
    gui.mode, gui.H, gui.W, gui.gui = "render", H, W, False
    gui.reference_skeleton, gui.buffer_overlay = None, None
    gui.skeleton_edge_img = np.zeros((H, W, 4), np.float32)
    gui.update_skeleton_edges = lambda camera, d_nodes: None
    gui.iteration = 20000
    torch.cuda.Event = lambda **k: MagicMock(elapsed_time=lambda e: 1.0)
    torch.cuda.synchronize = lambda *a, **k: None
    for mode in ("onlySkeleton", "Skeleton"):
        gui.visualization_mode = mode
        with S.quiet():
            gui.test_step(specified_cam=cams[1])
    gui.mode = "skinning"
    sys.modules["skeleton_utils.visualization"].get_color_for_skinning_weights = render_rig.get_color_for_skinning_weights
    gui.visualization_mode = "RGB"
    with S.quiet():
        gui.test_step(specified_cam=cams[2])
    with cite("interactive_GUI.py:265"):
        skeleton.deform.as_gaussians
    with cite("interactive_GUI.py:282"):
        skeleton.deform.node_radius
    with cite("interactive_GUI.py:913"):
        skeleton.deform.cached_nn_weight = not skeleton.deform.cached_nn_weight

    # a read that was already recorded from the same line with the same result (a loop reading ``deform.nodes`` 180 times)
    # is counted on its first occurrence instead of listed again
    packed, seen = [], {}
    for ev in log:
        key = json.dumps(ev, sort_keys=True) if ev["op"] == "get" else None
        if key is not None and key in seen:
            seen[key]["repeat"] = seen[key].get("repeat", 1) + 1
            continue
        ev = dict(ev)
        if key is not None:
            seen[key] = ev
        packed.append(ev)
    log = packed
    ctor_desc = {k: A.describe(v) for k, v in ctor.items()}
    out = {"seed": SEED, "constructor": ctor_desc,
           "optimization_params": {k: v for k, v in vars(opt).items() if isinstance(v, (bool, int, float))},
           "events": log}
    with open(os.path.join(HERE, "skeleton_api_calls.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    by = {}
    for ev in log:
        by[(ev["path"], ev["op"], ev["name"])] = by.get((ev["path"], ev["op"], ev["name"]), 0) + 1
    print("wrote skeleton_api_calls.json: %d events, %d distinct (path, op, name)" % (len(log), len(by)))
    for k in sorted(by):
        print("  %-9s %-8s %-28s x%d" % (k[0], k[1], k[2], by[k]))


if __name__ == "__main__":
    main()
