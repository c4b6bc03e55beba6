"""Parameter store with the attribute names / layouts / getters of the reference's
``GaussianModel`` (/root/reference/scene/gaussian_model.py:37-132, :177-195) — only what the
hot path reads.  Densification, optimizer surgery and PLY I/O are SURVEY.md §8-f items."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class GaussianModel:
    def __init__(self, sh_degree: int = 3, fea_dim: int = 0, with_motion_mask: bool = False,
                 use_isotropic_gs: bool = False):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.fea_dim = fea_dim + (1 if with_motion_mask else 0)
        self.with_motion_mask = with_motion_mask
        self.use_isotropic_gs = use_isotropic_gs
        e = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e
        self.feature = e
        self.max_radii2D = e

    @classmethod
    def from_tensors(cls, xyz, features_dc, features_rest, scaling, rotation, opacity, device="cuda",
                     use_isotropic_gs=False, active_sh_degree=3):
        gm = cls(3, use_isotropic_gs=use_isotropic_gs)
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

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1
