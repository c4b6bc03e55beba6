"""Golden vectors of the stage-2 OBJECTIVE (build container only — imports /root/reference):

    python tests/golden/make_objective_golden.py     ->  tests/golden/objective_tree8_n48.npz

The reference's own ``TrainRig.render_and_cal_loss`` (train_rig.py:416-515) runs, unmodified, on a ``TrainRig`` made with
``__new__`` (the scaffolding of record_api.py), once with the template camera (``viewpoint_cam.uid == template_idx``: the
template-offsets L2 is weighted x1e3, :446-456, and the ``template_fixed`` term on ``local_rotation`` is on, :474-482) and once
with another camera.  The rasterizer's source is absent from /root/reference (.gitmodules:1-6), so ``train_rig.render`` is
replaced by a differentiable stand-in — a fixed linear image of ``d_xyz`` squashed by a sigmoid — which gives the image term
(:508-514) a gradient w.r.t. ``d_xyz`` without touching what this fixture pins: the loss VALUE of every term the reference
logs, the total, and the gradients w.r.t. ``template_offsets``, ``local_rotation``, ``d_xyz`` and ``d_nodes``.
``lambda_deformed_node_prjection`` is 0 here (the projection term has its own goldens: skelproj_*.npz).

Only data leaves this script."""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import record_api as RA  # noqa: E402  (installs the shim, imports the reference's train_rig)

train_rig = RA.train_rig


def np_(t):
    return t.detach().cpu().numpy()


def main(name="objective_tree8_n48", seed=131, J=8, N=48, H=16, W=12):
    g = torch.Generator().manual_seed(seed)
    opt = RA.params(RA.OptimizationParams)
    dataset = RA.params(RA.ModelParams)
    pipe = RA.params(RA.PipelineParams)
    dataset.is_blender = True
    opt.lambda_deformed_node_prjection = 0.0
    A = (0.3 * torch.randn(3 * H * W, 3 * N, generator=g) / (3 * N) ** 0.5).half().float()  # (stored as fp16: exact)
    base = 0.2 * torch.randn(3 * H * W, generator=g)

    def render_standin(viewpoint_cam, gaussians, pipe_, bg, d_xyz, d_rotation, d_scaling, **kw):
        img = torch.sigmoid(A @ d_xyz.reshape(-1) + base).view(3, H, W)
        return {"render": img}
    train_rig.render = render_standin

    rig = train_rig.TrainRig.__new__(train_rig.TrainRig)
    rig.dataset, rig.opt, rig.pipe = dataset, opt, pipe
    rig.device = "cpu"
    rig.iteration = opt.optimize_template_offsets_iters + 10
    rig.template_idx = 2
    rig.background = torch.zeros(3)
    rig.gaussians = types.SimpleNamespace(use_isotropic_gs=False)
    deform = types.SimpleNamespace(use_template_offsets=True, template_offsets=None)
    rig.skeleton = types.SimpleNamespace(deform=deform, d_rot_as_res=True)
    res = dict(N=N, J=J, H=H, W=W, template_idx=rig.template_idx, A=np_(A).astype(np.float16), base=np_(base),
               lambda_template_offsets=float(opt.lambda_template_offsets), lambda_template_fixed=float(opt.lambda_template_fixed),
               lambda_rendering_image=float(opt.lambda_rendering_image), lambda_dssim=float(opt.lambda_dssim))
    T0 = 0.02 * torch.randn(N, 3, generator=g)
    q0 = torch.tensor([1.0, 0, 0, 0]) + 0.2 * torch.randn(J, 4, generator=g)
    dx0 = 0.1 * torch.randn(N, 3, generator=g)
    dn0 = 0.1 * torch.randn(J, 3, generator=g)
    gt_img = torch.rand(3, H, W, generator=g)
    res.update(template_offsets=np_(T0), local_rotation=np_(q0), d_xyz=np_(dx0), d_nodes=np_(dn0), gt_image=np_(gt_img))
    for tag, uid in (("template", rig.template_idx), ("other", 0)):
        T = T0.clone().requires_grad_(True)
        q = q0.clone().requires_grad_(True)
        dx = dx0.clone().requires_grad_(True)
        dn = dn0.clone().requires_grad_(True)
        deform.template_offsets = T
        rig.tb_writer = MagicMock()
        cam = types.SimpleNamespace(uid=uid, original_image=gt_img, gt_alpha_mask=None)
        d_values = {"d_xyz": dx, "d_rotation": torch.zeros(N, 4), "d_scaling": torch.zeros(N, 3), "d_opacity": None, "d_color": None,
                    "d_nodes": dn, "local_rotation": q}
        loss, pkg = rig.render_and_cal_loss(False, d_values, cam)
        loss.backward()
        scalars = {c.args[0].split("/")[-1]: float(c.args[1]) for c in rig.tb_writer.add_scalar.call_args_list}
        res[tag + "_uid"] = uid
        res[tag + "_loss"] = float(loss)
        for k, v in scalars.items():
            res["%s_%s" % (tag, k)] = v
        res[tag + "_g_template_offsets"] = np_(T.grad)
        res[tag + "_g_local_rotation"] = np_(q.grad) if q.grad is not None else np.zeros((J, 4), np.float32)
        res[tag + "_g_d_xyz"] = np_(dx.grad)
        res[tag + "_g_d_nodes"] = np_(dn.grad) if dn.grad is not None else np.zeros((J, 3), np.float32)
        res[tag + "_render"] = np_(pkg["render"])
        print(tag, "loss", float(loss), scalars)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
    print("wrote", name)


if __name__ == "__main__":
    main()
