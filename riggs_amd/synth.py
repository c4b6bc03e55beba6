"""Seeded synthetic scenes for parity tests and bench.py.

Everything is generated on the CPU with a ``torch.Generator`` and uploaded by the
caller, so that the CPU oracle and the HIP path see bit-identical inputs.  The
camera helpers restate the reference's matrix conventions
(/root/reference/utils/graphics_utils.py:42-100, scene/cameras.py:61-72): all 4x4
matrices are the transposed (row-vector) form that ``render()`` hands to the
rasterizer.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class Camera:
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # (4,4) transposed
    full_proj_transform: torch.Tensor   # (4,4) transposed
    camera_center: torch.Tensor         # (3,)
    fid: torch.Tensor

    def to(self, device):
        return Camera(self.image_height, self.image_width, self.FoVx, self.FoVy,
                      self.world_view_transform.to(device), self.full_proj_transform.to(device),
                      self.camera_center.to(device), self.fid.to(device))


def _projection(znear, zfar, fovX, fovY):
    tanY, tanX = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = tanY * znear, tanX * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def _projection_from_K(znear, zfar, K, W, H):
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    top, bottom = znear * cy / fy, -znear * (H - cy) / fy
    right, left = znear * (W - cx) / fx, -znear * cx / fx
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = -(right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_camera(H, W, azimuth_deg=45.0, elevation_deg=20.0, radius=4.0, fovx=0.6911112, fovy=None, K=None,
                   fid=0.37, znear=0.01, zfar=100.0) -> Camera:
    """D-NeRF-like orbit camera looking at the origin (+z forward, +y down)."""
    fovy = fovx if fovy is None else fovy
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    eye = np.array([radius * math.cos(el) * math.sin(az), -radius * math.sin(el), -radius * math.cos(el) * math.cos(az)])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(np.array([0.0, -1.0, 0.0]), fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    c2w_R = np.stack([right, up, fwd], axis=1)
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = c2w_R.T
    Rt[:3, 3] = -c2w_R.T @ eye
    Rt[3, 3] = 1.0
    wv = torch.tensor(np.float32(Rt)).transpose(0, 1)
    P = _projection_from_K(znear, zfar, K, W, H) if K is not None else _projection(znear, zfar, fovx, fovy)
    proj = P.transpose(0, 1)
    full = wv.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
    center = wv.inverse()[3, :3]
    return Camera(H, W, fovx, fovy, wv.contiguous(), full.contiguous(), center.contiguous(), torch.tensor([fid]))


def make_skeleton(g: torch.Generator, J: int, chain: bool = False):
    if chain:
        parents = torch.arange(-1, J - 1, dtype=torch.int64)
        joints = torch.stack([torch.zeros(J), -0.8 + 1.6 * torch.arange(J) / (J - 1), torch.zeros(J)], -1)
        joints = joints + 0.01 * torch.randn(J, 3, generator=g)
    else:
        parents = torch.full((J,), -1, dtype=torch.int64)
        joints = torch.zeros(J, 3)
        for i in range(1, J):
            parents[i] = int(torch.randint(0, i, (1,), generator=g))
            joints[i] = joints[parents[i]] + 0.25 * torch.randn(3, generator=g)
        joints = joints / joints.norm(dim=1).max()
    return joints.contiguous(), parents


def make_scene(N: int, J: int, seed: int, chain: bool = False, scale: float = 0.012, sh_rest_std: float = 0.1):
    """Gaussians scattered around the bones of a random skeleton."""
    g = torch.Generator().manual_seed(seed)
    joints, parents = make_skeleton(g, J, chain)
    bone = torch.randint(1, J, (N,), generator=g)
    t = torch.rand(N, 1, generator=g)
    a, b = joints[parents[bone]], joints[bone]
    xyz = a + t * (b - a) + 0.06 * torch.randn(N, 3, generator=g)
    scaling = math.log(scale) + 0.35 * torch.randn(N, 3, generator=g)
    q = torch.randn(N, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True) * (0.5 + torch.rand(N, 1, generator=g))
    opacity = 1.5 * torch.randn(N, 1, generator=g)
    f_dc = torch.randn(N, 1, 3, generator=g)
    f_rest = sh_rest_std * torch.randn(N, 15, 3, generator=g)
    rng = float(xyz.max() - xyz.min())
    node_radius = torch.full((J,), math.log(0.1 * rng))
    local_rot = torch.tensor([1.0, 0, 0, 0]) + 0.1 * torch.randn(J, 4, generator=g)
    global_trans = 0.02 * torch.randn(3, generator=g)
    return {
        "joints": joints, "parents": parents, "node_radius": node_radius,
        "xyz": xyz.contiguous(), "scaling": scaling, "rotation": q.contiguous(), "opacity": opacity,
        "features_dc": f_dc, "features_rest": f_rest,
        "local_rotation": local_rot, "global_trans": global_trans,
        "motion_mask": torch.ones(N, 1),
    }


def make_surface_scene(N: int, J: int, seed: int, shell: float = 0.09, scale: float = 0.006):
    """A dense-gradient variant of ``make_scene``: the Gaussians form a thin, mostly opaque SKIN around the bones (distance
    ``shell`` from the bone axis, opacity logit ~ N(2.5, 0.5)) instead of a deep semi-transparent cloud — a surface-like
    capture, where pixels saturate after a few layers and most FRONT-facing Gaussians receive a gradient.  Same
    skeleton, pose and parameter distributions otherwise."""
    sc = make_scene(N, J, seed, scale=scale)
    g = torch.Generator().manual_seed(seed + 77)
    joints, parents = sc["joints"], sc["parents"]
    bone = torch.randint(1, J, (N,), generator=g)
    t = torch.rand(N, 1, generator=g)
    a, b = joints[parents[bone]], joints[bone]
    axis = torch.nn.functional.normalize(b - a, dim=1)
    r = torch.randn(N, 3, generator=g)
    r = torch.nn.functional.normalize(r - (r * axis).sum(1, keepdim=True) * axis, dim=1)  # unit normal to the bone
    sc["xyz"] = (a + t * (b - a) + shell * (1.0 + 0.03 * torch.randn(N, 1, generator=g)) * r).contiguous()
    sc["opacity"] = 2.5 + 0.5 * torch.randn(N, 1, generator=g)
    return sc
