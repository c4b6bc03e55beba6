"""Parameter store with the class surface of the reference's ``GaussianModel`` / ``StandardGaussianModel``
(/root/reference/scene/gaussian_model.py:37-546): attribute names and layouts, getters, ``create_from_pcd`` (the one
caller of ``distCUDA2``), ``training_setup`` and the learning-rate hooks, ``get_covariance``, PLY I/O, and the
densification surgery — the row work of which runs on the device (csrc/densify.hip) and the optimizer step in one HIP
launch (``riggs_amd.optim.FusedAdam``).  ``SkeletonWarp.as_gaussians`` builds its J-point model from this class when the
trainer's own ``scene.gaussian_model`` is not importable."""
from __future__ import annotations

import math
from typing import NamedTuple

import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicPointCloud(NamedTuple):  # utils/graphics_utils.py:17-20
    points: object
    colors: object
    normals: object


_SH_C0 = 0.28209479177387814


def RGB2SH(rgb):  # utils/sh_utils.py:115-116
    return (rgb - 0.5) / _SH_C0


def inverse_sigmoid(x):  # utils/general_utils.py:24-25
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation lr_init -> lr_final over ``max_steps`` with the optional sine warm-up of the first
    ``lr_delay_steps`` (utils/general_utils.py:49-82); negative steps or two zero rates switch the group off."""
    def rate(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        warm = 1.0
        if lr_delay_steps > 0:
            warm = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        t = min(max(step / max_steps, 0.0), 1.0)
        return warm * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return rate


def quaternion_multiply(a, b):  # Hamilton product, (w, x, y, z): scene/gaussian_model.py:25-34
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def build_rotation(r):
    """(N, 4) quaternions (w, x, y, z), normalised here, -> (N, 3, 3) (utils/general_utils.py:137-160)."""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), -1).reshape(-1, 3, 3)


def build_scaling_rotation(s, r):  # R diag(s): utils/general_utils.py:163-172
    return build_rotation(r) * s[:, None, :]


def strip_symmetric(m):  # upper triangle xx xy xz yy yz zz: utils/general_utils.py:121-134
    return torch.stack((m[:, 0, 0], m[:, 0, 1], m[:, 0, 2], m[:, 1, 1], m[:, 1, 2], m[:, 2, 2]), -1)


def farthest_point_sample(xyz, npoint):
    """(B, N, C) -> (B, npoint) indices of an iterative farthest-point sweep from a random start
    (utils/time_utils.py:461-482)."""
    B, N, _ = xyz.shape
    dev = xyz.device
    picked = torch.zeros(B, npoint, dtype=torch.long, device=dev)
    nearest = torch.full((B, N), 1e10, device=dev)
    cur = torch.randint(0, N, (B,), dtype=torch.long, device=dev)
    rows = torch.arange(B, device=dev)
    for i in range(npoint):
        picked[:, i] = cur
        d = ((xyz - xyz[rows, cur][:, None]) ** 2).sum(-1)
        nearest = torch.minimum(nearest, d)
        cur = nearest.argmax(-1)
    return picked


class GaussianModel:
    def __init__(self, sh_degree: int, fea_dim=0, with_motion_mask=True, use_isotropic_gs=False, **kwargs):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.with_motion_mask = with_motion_mask
        self.fea_dim = fea_dim + (1 if with_motion_mask else 0)  # the mask is the last feature column (:57-61)
        self.use_isotropic_gs = use_isotropic_gs
        e = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e
        self.feature = e
        self.max_radii2D = e
        self.xyz_gradient_accum = e
        self.optimizer = None
        self.scaling_activation, self.scaling_inverse_activation = torch.exp, torch.log
        self.opacity_activation, self.inverse_opacity_activation = torch.sigmoid, inverse_sigmoid
        self.rotation_activation = F.normalize
        self.covariance_activation = self._covariance_from_scaling_rotation

    @staticmethod
    def _covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):  # :40-44
        M = build_scaling_rotation(scaling_modifier * scaling, rotation)
        return strip_symmetric(M @ M.transpose(1, 2))

    def param_names(self):  # :81-82
        return ["_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "max_radii2D", "xyz_gradient_accum"]

    @classmethod
    def build_from(cls, gs, **kwargs):  # :84-95 (shares the geometry, zeroes the colours)
        new = GaussianModel(**kwargs)
        new._xyz, new._scaling, new._rotation = nn.Parameter(gs._xyz), nn.Parameter(gs._scaling), nn.Parameter(gs._rotation)
        new._features_dc = nn.Parameter(torch.zeros_like(gs._features_dc))
        new._features_rest = nn.Parameter(torch.zeros_like(gs._features_rest))
        new._opacity, new.feature = nn.Parameter(gs._opacity), nn.Parameter(gs.feature)
        new.max_radii2D = torch.zeros(new.get_xyz.shape[0], device=gs._xyz.device)
        return new

    @classmethod
    def from_tensors(cls, xyz, features_dc, features_rest, scaling, rotation, opacity, device="cuda",
                     use_isotropic_gs=False, active_sh_degree=3):
        gm = cls(3, with_motion_mask=False, use_isotropic_gs=use_isotropic_gs)
        P = lambda t: nn.Parameter(t.detach().to(device).float().contiguous().requires_grad_(True))  # noqa: E731
        gm._xyz, gm._features_dc, gm._features_rest = P(xyz), P(features_dc), P(features_rest)
        gm._scaling, gm._rotation, gm._opacity = P(scaling), P(rotation), P(opacity)
        gm.active_sh_degree = active_sh_degree
        gm.max_radii2D = torch.zeros(xyz.shape[0], device=device)
        return gm

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    @property
    def motion_mask(self):  # gaussian_model.py:97-102
        if self.with_motion_mask:
            return torch.sigmoid(self.feature[..., -1:])
        m = getattr(self, "_ones_mask", None)
        if m is None or m.shape[0] != self._xyz.shape[0] or m.device != self._xyz.device:
            m = self._ones_mask = torch.ones_like(self._xyz[..., :1])  # constant when with_motion_mask is off
        return m

    @property
    def get_scaling(self):  # :104-110
        if self.use_isotropic_gs:
            return torch.exp(self._scaling[..., :1].repeat(1, 3))
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return F.normalize(self._rotation)

    def get_rotation_bias(self, rotation_bias=None):  # :116-118
        rotation_bias = rotation_bias if rotation_bias is not None else 0.0
        return F.normalize(self._rotation + rotation_bias)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):  # :124-128
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier=1, d_rotation=None, gs_rot_bias=None):
        """(N, 6) upper triangle of R S S^T R^T; a rotation residual composes by quaternion PRODUCT here — not the
        additive residual of the default branch (:134-142)."""
        rotation = self._rotation if d_rotation is None else quaternion_multiply(self._rotation, d_rotation)
        if gs_rot_bias is not None:
            rotation = quaternion_multiply(gs_rot_bias, rotation / rotation.norm(dim=-1, keepdim=True))
        return self.covariance_activation(self.get_scaling, scaling_modifier, rotation)

    def get_covariance_inv(self):  # :144-147
        M = build_rotation(self._rotation).transpose(1, 2) * (1.0 / self.get_scaling)[:, None, :]
        return M @ M.transpose(1, 2)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def create_from_pcd(self, pcd, spatial_lr_scale: float = 5.0, print_info=True, max_point_num=150_000):
        """Initial cloud from points + colours (:153-195): DC coefficients from the colours, isotropic log-scales from
        the mean squared distance to the 3 nearest neighbours (``distCUDA2``: csrc/knn.hip), identity rotations,
        opacity 0.1, features -1e-2 (mask column 0).  Arrays go to ``cuda``; tensors stay on their device."""
        import numpy as np
        from .knn import distCUDA2
        self.spatial_lr_scale = 1
        pts = torch.tensor(np.asarray(pcd.points)).float().cuda() if isinstance(pcd.points, np.ndarray) else pcd.points
        col = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().cuda()) if isinstance(pcd.colors, np.ndarray) else pcd.colors
        n, dev = pts.shape[0], pts.device
        if print_info:
            print("Number of points at initialisation : ", n)
        n_coef = (self.max_sh_degree + 1) ** 2
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None]
        if not self.use_isotropic_gs:
            scales = scales.repeat(1, 3)
        rots = torch.zeros((n, 4), device=dev)
        rots[:, 0] = 1
        P = lambda t: nn.Parameter(t.contiguous().requires_grad_(True))  # noqa: E731
        self._xyz = P(pts)
        self._features_dc = P(col.to(dev).float().reshape(n, 1, 3).clone())
        self._features_rest = P(torch.zeros((n, n_coef - 1, 3), device=dev))
        self._scaling, self._rotation = P(scales), P(rots)
        self._opacity = P(inverse_sigmoid(0.1 * torch.ones((n, 1), dtype=torch.float, device=dev)))
        self.max_radii2D = torch.zeros(n, device=dev)
        self.feature = nn.Parameter(-1e-2 * torch.ones((n, self.fea_dim), dtype=torch.float32, device=dev), requires_grad=True)
        if self.with_motion_mask:
            self.feature.data[..., -1] = 0.0
        self._ones_mask = None

    # ---- optimizer (scene/gaussian_model.py:197-231, :516-518) --------------------------------------------------
    def training_setup(self, training_args, capturable=False):
        """Same groups, names and learning rates as the reference; the optimizer is ``riggs_amd.optim.FusedAdam``
        (a torch.optim.Adam whose step is one HIP launch)."""
        from .optim import FusedAdam
        self.percent_dense = training_args.percent_dense
        n, dev = self.get_xyz.shape[0], self.get_xyz.device
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.spatial_lr_scale = 5
        groups = [
            {"params": [self._xyz], "lr": training_args.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": training_args.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": training_args.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": training_args.scaling_lr * self.spatial_lr_scale, "name": "scaling"},
            {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
        ]
        if self.fea_dim > 0:
            groups.append({"params": [self.feature], "lr": training_args.feature_lr, "name": "feature"})
        if capturable:  # scheduled learning rate as a device scalar (see FusedAdam): update_learning_rate fills it
            groups[0]["lr"] = torch.tensor(float(groups[0]["lr"]), dtype=torch.float32, device=dev)
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15, capturable=capturable)
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=training_args.position_lr_init * self.spatial_lr_scale,
                                                    lr_final=training_args.position_lr_final * self.spatial_lr_scale,
                                                    lr_delay_mult=training_args.position_lr_delay_mult,
                                                    max_steps=training_args.position_lr_max_steps)
        self.skeleton_gs_position_lr = getattr(training_args, "skeleton_gs_position_lr",
                                               training_args.position_lr_init * self.spatial_lr_scale)

    def _set_xyz_lr(self, lr):
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                if isinstance(group["lr"], torch.Tensor):
                    group["lr"].fill_(lr)
                else:
                    group["lr"] = lr

    def update_learning_rate(self, iteration):  # :222-228
        self._set_xyz_lr(self.xyz_scheduler_args(iteration))

    def update_learning_rate_for_skeleton_step(self, iteration):  # :231-235
        self._set_xyz_lr(self.skeleton_gs_position_lr)

    def add_densification_stats(self, viewspace_point_tensor, update_filter, radii=None):  # :516-518 (+ train_rig.py:333-335)
        from .optim import densify_stats
        densify_stats(viewspace_point_tensor.grad, update_filter, self.xyz_gradient_accum, self.denom, radii,
                      self.max_radii2D if radii is not None else None)

    # ---- checkpoints (scene/gaussian_model.py:232-336) ------------------------------------------------------------
    def construct_list_of_attributes(self):  # :232-251
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += ["f_dc_%d" % i for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += ["f_rest_%d" % i for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names.append("opacity")
        names += ["scale_%d" % i for i in range(self._scaling.shape[1])]
        names += ["rot_%d" % i for i in range(self._rotation.shape[1])]
        if self.fea_dim > 0:
            names += ["fea_%d" % i for i in range(self.feature.shape[1])]
        return names

    def save_ply(self, path):  # :253-272 (same column order and (coefficient-major) SH flattening)
        import os

        import numpy as np

        from .ply import write_vertex_ply
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        c = lambda t: t.detach().cpu().numpy()  # noqa: E731
        xyz = c(self._xyz)
        cols = [xyz, np.zeros_like(xyz), c(self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous()),
                c(self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous()), c(self._opacity),
                c(self._scaling), c(self._rotation)]
        if self.fea_dim > 0:
            cols.append(c(self.feature))
        write_vertex_ply(path, self.construct_list_of_attributes(), np.concatenate(cols, axis=1))

    def load_ply(self, path, og_number_points=-1, device="cuda"):  # :279-336
        import numpy as np

        from .ply import read_vertex_ply
        self.og_number_points = og_number_points
        names, v = read_vertex_ply(path)
        col = lambda n: np.asarray(v[n], dtype=np.float32)  # noqa: E731
        xyz = np.stack((col("x"), col("y"), col("z")), axis=1)
        n = xyz.shape[0]
        dc = np.stack((col("f_dc_0"), col("f_dc_1"), col("f_dc_2")), axis=1)[:, :, None]
        rest_names = [k for k in names if k.startswith("f_rest_")]
        assert len(rest_names) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        rest = np.stack([col(k) for k in rest_names], axis=1).reshape(n, 3, (self.max_sh_degree + 1) ** 2 - 1)
        scales = np.stack([col(k) for k in names if k.startswith("scale_")], axis=1)
        if self.use_isotropic_gs:
            scales = scales[..., :1]
        rots = np.stack([col(k) for k in names if k.startswith("rot")], axis=1)
        feas = np.zeros((n, self.fea_dim), np.float32)
        for i, k in enumerate([k for k in names if k.startswith("fea")]):
            feas[:, i] = col(k)
        P = lambda a: nn.Parameter(torch.tensor(a, dtype=torch.float, device=device).contiguous().requires_grad_(True))  # noqa: E731
        self._xyz = P(xyz)
        self._features_dc = nn.Parameter(torch.tensor(dc, dtype=torch.float, device=device).transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(torch.tensor(rest, dtype=torch.float, device=device).transpose(1, 2).contiguous().requires_grad_(True))
        self._opacity = P(col("opacity")[:, None])
        self._scaling = P(scales)
        self._rotation = P(rots)
        if self.fea_dim > 0:
            self.feature = P(feas)
        self.max_radii2D = torch.zeros(n, device=device)
        self.active_sh_degree = self.max_sh_degree

    # ---- densification / pruning (scene/gaussian_model.py:275-278, 338-514; train_rig.py:359-365) --------------------------
    # Same method names, arguments and results as the reference.  The row work — every parameter tensor and both Adam moments
    # of every group re-assembled for the new cloud — is ONE gather launch from an index plan (csrc/densify.hip); the plan comes
    # from stream compactions of byte flags, and densify_and_prune's three predicates from one selection kernel.
    def _groups(self):
        return {g["name"]: g for g in self.optimizer.param_groups}

    def _compact(self, *flag_vectors):
        """Ascending int32 indices of the non-zero entries of uint8 / bool device vectors (riggs_compact_indices): one launch
        per vector, ONE device->host read for all the counts.  Returns a tensor per vector (the tensor itself for one)."""
        from . import _lib as L
        lib = L.lib()
        cnts = torch.empty(len(flag_vectors), dtype=torch.int32, device=flag_vectors[0].device)
        out = []
        for k, flags in enumerate(flag_vectors):
            f = flags.to(torch.uint8).contiguous()
            n, dev = f.numel(), f.device
            idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
            ws = torch.empty(lib.riggs_compact_workspace_bytes(n), dtype=torch.uint8, device=dev)
            L.check(lib.riggs_compact_indices(n, f.data_ptr(), idx.data_ptr(), cnts[k:].data_ptr(), ws.data_ptr(), L.stream_ptr()),
                    "riggs_compact_indices")
            out.append(idx)
        out = [idx[:c] for idx, c in zip(out, cnts.tolist())]
        return out[0] if len(out) == 1 else out

    def _regather(self, plan, stats="zero", children=None):
        """The new cloud from an index plan: row m of every tensor comes from row ``plan[m]`` of the old one; ``plan[m] < 0``
        marks a NEW row made from source ``~plan[m]`` (its Adam moments start at zero: cat_tensors_to_optimizer :394-417; kept
        rows keep theirs: _prune_optimizer :355-371).  ``children`` = (first row, parents (S,), copies, unit normals or None):
        the rows of split children, whose position and scale are then re-drawn (densify_and_split :452-460).  ``stats``:
        "zero" (densification_postfix :435-437) or "gather" (prune_points :388-392)."""
        import ctypes as C
        from . import _lib as L
        lib = L.lib()
        plan = plan.to(torch.int32).contiguous()
        n_out, dev = plan.numel(), plan.device
        src, dst, width, zero_new, owners = [], [], [], [], []
        for g in self.optimizer.param_groups:
            p = g["params"][0]
            st = self.optimizer.state.get(p, None)
            new_p = torch.empty((n_out,) + tuple(p.shape[1:]), dtype=torch.float32, device=dev)
            src.append(p.detach().contiguous()); dst.append(new_p); width.append(max(1, p.numel() // max(1, p.shape[0])) if p.shape[0] else int(math.prod(p.shape[1:]))); zero_new.append(0)
            moments = None
            if st is not None and "exp_avg" in st:
                moments = (torch.empty_like(new_p), torch.empty_like(new_p))
                for m_old, m_new in zip((st["exp_avg"], st["exp_avg_sq"]), moments):
                    src.append(m_old.contiguous()); dst.append(m_new); width.append(width[-1]); zero_new.append(1)
            owners.append((g, p, st, new_p, moments))
        if stats == "gather":
            new_stats = []
            for t in (self.xyz_gradient_accum, self.denom, self.max_radii2D):
                nt = torch.empty((n_out,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev)
                src.append(t.contiguous()); dst.append(nt); width.append(1); zero_new.append(1)
                new_stats.append(nt)
        k = len(src)
        if n_out:
            L.check(lib.riggs_rows_gather(n_out, plan.data_ptr(), k, (C.c_void_p * k)(*[t.data_ptr() for t in src]),
                                          (C.c_void_p * k)(*[t.data_ptr() for t in dst]), (C.c_int32 * k)(*width),
                                          (C.c_uint8 * k)(*zero_new), L.stream_ptr()), "riggs_rows_gather")
        if children is not None and children[1].numel():
            row0, parents, copies, z = children
            S = int(parents.numel())
            n_ch = S * copies
            if z is None:
                z = torch.randn(n_ch, 3, device=dev)
            z = z.to(dev, torch.float32).contiguous()
            grp = self._groups()
            old_xyz, old_sc, old_rot = grp["xyz"]["params"][0], grp["scaling"]["params"][0], grp["rotation"]["params"][0]
            new_xyz = next(o[3] for o in owners if o[0]["name"] == "xyz")
            new_sc = next(o[3] for o in owners if o[0]["name"] == "scaling")
            cols = old_sc.shape[1]
            L.check(lib.riggs_split_children(n_ch, S, cols, parents.data_ptr(), z.data_ptr(), old_xyz.data_ptr(), old_sc.data_ptr(),
                                             old_rot.data_ptr(), 0.8 * copies, new_xyz[row0:].data_ptr(), new_sc[row0:].data_ptr(),
                                             L.stream_ptr()), "riggs_split_children")
        out = {}
        for g, p, st, new_p, moments in owners:
            new_param = nn.Parameter(new_p.requires_grad_(True))
            if st is not None:
                del self.optimizer.state[p]
                if moments is not None:
                    st["exp_avg"], st["exp_avg_sq"] = moments
                self.optimizer.state[new_param] = st
            g["params"][0] = new_param
            out[g["name"]] = new_param
        self._drop_launch_plan()
        self._xyz, self._features_dc, self._features_rest = out["xyz"], out["f_dc"], out["f_rest"]
        self._opacity, self._scaling, self._rotation = out["opacity"], out["scaling"], out["rotation"]
        if self.fea_dim > 0:
            self.feature = out["feature"]
        if stats == "gather":
            self.xyz_gradient_accum, self.denom, self.max_radii2D = new_stats
        else:
            self.xyz_gradient_accum = torch.zeros((n_out, 1), device=dev)
            self.denom = torch.zeros((n_out, 1), device=dev)
            self.max_radii2D = torch.zeros(n_out, device=dev)
        self._ones_mask = None
        return out

    def _drop_launch_plan(self):
        """The optimizer's cached launch plan (riggs_amd.optim._Plan) holds the OLD parameter objects, their moments and the last
        step's gradients: after surgery installed new tensors it would keep the old full-size ones alive through the whole next
        forward / backward (it is rebuilt by the next step anyway)."""
        if self.optimizer is not None and hasattr(self.optimizer, "_hip_plan"):
            self.optimizer._hip_plan = None

    def replace_tensor_to_optimizer(self, tensor, name):  # :338-353
        self._drop_launch_plan()
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] == name:
                st = self.optimizer.state.get(group["params"][0], None)
                new_param = nn.Parameter(tensor.requires_grad_(True))
                if st is not None:
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(tensor), torch.zeros_like(tensor)
                    del self.optimizer.state[group["params"][0]]
                    self.optimizer.state[new_param] = st
                group["params"][0] = new_param
                out[name] = new_param
        return out

    def reset_opacity(self):  # :275-278
        op = torch.minimum(self.get_opacity, torch.full_like(self._opacity, 0.01))
        self._opacity = self.replace_tensor_to_optimizer(torch.log(op / (1 - op)), "opacity")["opacity"]

    def prune_points(self, mask):  # :373-392
        self._regather(self._compact(~mask.reshape(-1).bool()), stats="gather")

    def densify_and_clone(self, grads=None, grad_threshold=None, scene_extent=None, selected_pts_mask=None):  # :475-498
        if selected_pts_mask is None:
            selected_pts_mask = (torch.norm(grads, dim=-1) >= grad_threshold) & \
                (torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        n = self._xyz.shape[0]
        clones = self._compact(selected_pts_mask.reshape(-1))
        self._regather(torch.cat([torch.arange(n, dtype=torch.int32, device=clones.device), ~clones]), stats="zero")

    def densify_and_split(self, grads=None, grad_threshold=None, scene_extent=None, N=2, selected_pts_mask=None, without_prune=False,
                          unit_normals=None):  # :440-473
        """``unit_normals`` (S * N, 3): the standard-normal draws behind the children's offsets (the reference draws
        ``torch.normal(0, stds)`` = stds * z itself; tests hand the same z to both sides)."""
        n = self._xyz.shape[0]
        if selected_pts_mask is None:
            padded = torch.zeros(n, device=self._xyz.device)
            padded[:grads.shape[0]] = grads.squeeze()
            selected_pts_mask = (padded >= grad_threshold) & (torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        sel = selected_pts_mask.reshape(-1).bool()
        if without_prune:
            parents = self._compact(sel)
            kept = torch.arange(n, dtype=torch.int32, device=parents.device)
        else:
            parents, kept = self._compact(sel, ~sel)
        plan = torch.cat([kept] + [~parents] * N)
        self._regather(plan, stats="zero", children=(int(kept.numel()), parents, N, unit_normals))

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, unit_normals=None):  # :500-514
        """clone -> split -> prune as ONE selection + three compactions + ONE gather; the resulting rows and their order are
        the reference's: surviving old rows, surviving clones, surviving children (copy-major)."""
        from . import _lib as L
        n, dev = self._xyz.shape[0], self._xyz.device
        flags = torch.empty(3, n, dtype=torch.uint8, device=dev)
        sc = self._scaling.detach().contiguous()
        L.check(L.lib().riggs_densify_select(n, sc.shape[1], self.xyz_gradient_accum.contiguous().data_ptr(), self.denom.contiguous().data_ptr(),
                                             sc.data_ptr(), self._opacity.detach().contiguous().data_ptr(), float(max_grad),
                                             float(self.percent_dense * extent), float(min_opacity),
                                             float(0.1 * extent) if max_screen_size else -1.0, 1.6, flags.data_ptr(), L.stream_ptr()),
                "riggs_densify_select")
        kept, clones, parents = self._compact(flags[0], flags[1], flags[2])
        plan = torch.cat([kept, ~clones, ~parents, ~parents])
        self._regather(plan, stats="zero", children=(int(kept.numel() + clones.numel()), parents, 2, unit_normals))

    def sampling_and_prune(self, num_sample=5000):  # :520-531 (train_rig.py:196)
        keep = farthest_point_sample(self.get_xyz.detach()[None], num_sample)[0]
        n = self.get_xyz.shape[0]
        drop = torch.ones(n, dtype=torch.bool, device=self.get_xyz.device)
        drop[keep] = False
        if self.max_radii2D.shape[0] != n:
            self.max_radii2D = torch.zeros(n, device=self.get_xyz.device)
        self.prune_points(drop)
        print("sample gaussians from", n, "to", self.get_xyz.shape[0])


class StandardGaussianModel(GaussianModel):
    """Every axis (``all_the_same``: every Gaussian) shares the mean log-scale (:534-546) — the J-point model the
    skeleton's joints are drawn with (``SkeletonWarp.as_gaussians``)."""

    def __init__(self, sh_degree: int, fea_dim=0, with_motion_mask=True, all_the_same=False):
        super().__init__(sh_degree, fea_dim, with_motion_mask)
        self.all_the_same = all_the_same

    @property
    def get_scaling(self):
        s = self._scaling[..., :1].repeat(1, 3) if self.use_isotropic_gs else self._scaling
        mean = s.mean()[None, None] if self.all_the_same else s.mean(dim=1, keepdim=True)
        return self.scaling_activation(mean.expand_as(s))
